"""The fused step reads the caller's batch in place (HipTrainStep._bind_batch: the stems' plane tables and the ops that take the input /
target pointer are re-pointed at contiguous fp32 tensors of the plan's shapes) -- same bits as with the copies into the plan's static
buffers, across steps with different batches (re-binding), for both networks of the path, and with a fallback for anything else."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(arch, zero_copy, monkeypatch, odd_input=False):
    from radar_depth_amd import main as hmain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    monkeypatch.setenv("RD_ZERO_COPY_INPUT", "1" if zero_copy else "0")
    b, h, w = 2, 97, 161
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    made = hmain.create_model(args, [h, w])
    m, lw = made if isinstance(made, tuple) else (made, None)
    procedural_fill_(m)
    m = m.cuda().train()
    ts = HipTrainStep(m, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=lw)
    losses = []
    keep = []
    for k in range(3):
        x, t = make_batch(b, h, w, 100 + k)
        x, t = x.cuda(), t.cuda()
        if odd_input and k == 1:
            x = torch.cat([x, x[:, :1]], 1)[:, :4]            # a non-contiguous view: must take the copy path
            assert not x.is_contiguous()
        keep += [x, t]
        loss, _ = ts.step(x, t)
        losses.append(loss.clone())
    torch.cuda.synchronize()
    bound = ts._bound
    return [v.item() for v in losses], [p.detach().clone() for p in m.parameters()], bound, keep, ts


@pytest.mark.parametrize("arch", ["resnet18_latefusion", "resnet18_multistage_uncertainty_fixs"])
def test_step_reads_the_callers_batch_in_place(arch, monkeypatch):
    l0, p0, b0, _, ts0 = _run(arch, False, monkeypatch)
    l1, p1, b1, keep, ts1 = _run(arch, True, monkeypatch)
    assert b0 == [ts0.plan.x_in.data_ptr(), ts0.target.data_ptr()]              # copies: the ops point at the static buffers
    assert b1 == [keep[-2].data_ptr(), keep[-1].data_ptr()]                     # in place: at the last batch
    assert l0 == l1, (l0, l1)
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))


def test_noncontiguous_batch_takes_the_copy(monkeypatch):
    l0, p0, _, _, _ = _run("resnet18_latefusion", False, monkeypatch)
    l1, p1, b1, _, ts1 = _run("resnet18_latefusion", True, monkeypatch, odd_input=True)
    assert l0 == l1
    assert all(torch.equal(a, b) for a, b in zip(p0, p1))


def test_eager_forward_after_an_in_place_step_reads_its_own_input(monkeypatch):
    """A fused step with operands="fp32" and the eager nn.Module forward share one cached plan: after a step that pointed the stems at the
    caller's batch, the eager forward must read ITS input (run_forward re-binds the plan's own buffer), and the next step re-binds again."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    monkeypatch.setenv("RD_ZERO_COPY_INPUT", "1")
    b, h, w = 2, 97, 161
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    m = m.cuda().train()
    xa, ta = [v.cuda() for v in make_batch(b, h, w, 1)]
    xb, tb = [v.cuda() for v in make_batch(b, h, w, 2)]
    with torch.no_grad():
        want_b = m(xb).clone()                         # eager forward (training-mode statistics), before any step
    ts = HipTrainStep(m, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0, operands="fp32")     # lr = 0: the parameters stay put
    # (BatchNorm running statistics move, the training-mode forward does not read them)
    l1, _ = ts.step(xa, ta)
    l1 = l1.item()
    assert ts.plan._x_bound == xa.data_ptr()
    with torch.no_grad():
        got_b = m(xb).clone()
    torch.cuda.synchronize()
    assert torch.equal(got_b, want_b)
    l2, _ = ts.step(xa, ta)
    torch.cuda.synchronize()
    assert ts.plan._x_bound == xa.data_ptr() and l2.item() == l1


@pytest.mark.parametrize("arch,lanes", [("resnet18_latefusion", 3), ("resnet18_multistage_uncertainty_fixs", 3), ("resnet18_latefusion", 2)])
def test_multithreaded_issue_is_bit_identical(monkeypatch, arch, lanes):
    """RD_ISSUE_THREADS=n: the step's ops are issued from n host threads, one per plan stream (rd_optable_run_mt; every
    rd_stream_wait_event behind its rd_event_record).  Same launches in the same per-stream order: losses and parameters bit for bit
    those of the single-threaded issue, over several steps."""
    import types
    from radar_depth_amd.main import HipTrainStep, create_model
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    b, h, w = 2, 97, 161
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)

    def build():
        torch.manual_seed(0)
        made = create_model(args, [h, w])
        m, lw = made if isinstance(made, tuple) else (made, None)
        procedural_fill_(m)
        return m.cuda(), lw
    (m1, lw1), (m2, lw2) = build(), build()
    monkeypatch.setenv("RD_ISSUE_THREADS", "1")
    t1 = HipTrainStep(m1, b, h, w, loss_weights=lw1)
    monkeypatch.setenv("RD_ISSUE_THREADS", str(lanes))
    t2 = HipTrainStep(m2, b, h, w, loss_weights=lw2)
    assert t1._issue_threads == 1 and t2._issue_threads == lanes
    for it in range(4):
        x, t = [v.cuda() for v in make_batch(b, h, w, 40 + it, ref_pixels=h * w)]
        l1, _ = t1.step(x, t)
        l2, _ = t2.step(x, t)
        torch.cuda.synchronize()
        assert l1.item() == l2.item(), (it, l1.item(), l2.item())
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(p, q)
