"""GPU parity of rd_gconv_split (csrc/gconv_split.hip: fp32 convolution rebuilt from six bf16 MFMAs per product over three-piece
operands) against torch CPU fp32 convolutions, through the C ABI -- at the SAME tolerance as the fp32-MFMA kernel's tests
(tests/test_gpu_gconv.py: 2e-5 of the output's max magnitude) -- and against an fp64 convolution side by side with rd_gconv."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _plan_every_shape():
    """Plan every shape the kernel can run, also those the library leaves to rd_gconv because they measured slower there -- for the
    tests of THIS file only (rd_gconv_split_plan_all is a per-process switch of the library, restored behind every test)."""
    from radar_depth_amd._lib import lib
    prev = lib().rd_gconv_split_plan_all(1)
    yield
    lib().rd_gconv_split_plan_all(1 if prev == 1 else 0)


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


FWD = [
    (2, 64, 64, 3, 1, 1, 113, 200),   # layer1
    (2, 128, 128, 3, 1, 1, 57, 100),
    (2, 64, 128, 1, 2, 0, 113, 200),  # downsample
    (2, 64, 128, 3, 2, 1, 113, 200),  # stride 2: wide patch
    (2, 256, 256, 3, 1, 1, 29, 50),
    (2, 512, 512, 3, 1, 1, 15, 25),   # layer4
    (2, 640, 512, 1, 1, 0, 15, 25),   # conv_fusion
    (2, 32, 32, 3, 1, 1, 120, 200),   # decoder.layer3 conv2
    (3, 32, 48, 3, 1, 1, 9, 7),       # tiny / ragged
    (16, 512, 256, 1, 1, 0, 15, 25),
    (2, 48, 80, 3, 1, 1, 31, 17),     # channels that do not fill a block, 48-channel reduction
    (1, 96, 36, 3, 2, 1, 33, 45),
    (3, 64, 64, 3, 1, 1, 1, 1),
    (2, 32, 32, 3, 1, 1, 40, 1),
    (2, 80, 48, 1, 1, 0, 19, 23),
]


@pytest.mark.parametrize("cfg", FWD)
def test_gconv_split_forward(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
    if not ops.gconv_split_supported(d):
        pytest.skip("no split plan for this descriptor (callers keep rd_gconv)")
    xg = ops.nchw_to_nhwc(x.cuda())
    wp = ops.pack_weights_split(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
    stat = torch.zeros(ops.gconv_split_stat_tiles(d), 2, co, device="cuda")
    ops.gconv_split(d, xg, wp, out, stat=stat)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, y) < 2e-5, (_rel(got, y), cfg)
    s_ = stat.sum(0).cpu().double()
    ref_s = y.double().sum((0, 2, 3))
    ref_q = (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - ref_s).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert _rel(s_[1], ref_q) < 1e-4


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 3, 1, 1, 113, 200),
    (2, 64, 128, 3, 2, 1, 113, 200),
    (2, 64, 128, 1, 2, 0, 113, 200),
    (2, 256, 512, 3, 2, 1, 29, 50),
    (16, 512, 512, 3, 1, 1, 15, 25),
    (2, 48, 80, 3, 1, 1, 31, 17),
    (1, 96, 48, 3, 2, 1, 33, 45),
    (3, 64, 64, 3, 1, 1, 1, 1),
])
def test_gconv_split_dgrad(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * co)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
    if not ops.gconv_split_supported(d):
        pytest.skip("no split plan for this descriptor")
    wp = ops.pack_weights_split(wt.cuda(), transpose=True)
    dy = ops.nchw_to_nhwc(gy.cuda())
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    if zero_fill:
        ops.fill(dx, 0.0)
    add = torch.randn(n, h, w, ci, generator=g)
    ops.gconv_split(d, dy, wp, dx, addend=add.cuda() if not zero_fill else None, ld_add=ci)
    torch.cuda.synchronize()
    want = x.grad + (add.permute(0, 3, 1, 2) if not zero_fill else 0)
    got = dx.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, want) < 2e-5


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (32, 13, 9)])
def test_gconv_split_upproj(c, h, w):
    """UpProj forward (four parity phases of 9/6/6/4 taps, two weight tensors side by side) and its input gradient."""
    from radar_depth_amd import convdesc as cd, ops
    n = 2
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    w_up = torch.randn(c // 2, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    w_bt = torch.randn(c // 2, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    u = torch.zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x.detach()
    u.requires_grad_(True)
    wcat = torch.cat([w_up, w_bt], 0)
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.upproj_fwd(n, h, w, c, c)
    assert ops.gconv_split_supported(d)
    wp = ops.pack_weights_split(wcat.cuda())
    out = torch.full((n, 2 * h, 2 * w, c), float("nan"), device="cuda")
    ops.gconv_split(d, ops.nchw_to_nhwc(x.detach().cuda()), wp, out)
    torch.cuda.synchronize()
    assert _rel(out.permute(0, 3, 1, 2).cpu(), y.detach()) < 2e-5
    dd = cd.upproj_dgrad(n, h, w, c, c)
    if not ops.gconv_split_supported(dd):        # (stride-2 input: the patch of a tile can exceed what the staging waves hold)
        return
    wd = ops.pack_weights_split(wcat.cuda(), transpose=True)
    dx = torch.full((n, h, w, c), float("nan"), device="cuda")
    ops.gconv_split(dd, ops.nchw_to_nhwc(gy.cuda()), wd, dx)
    torch.cuda.synchronize()
    want = u.grad[:, :, ::2, ::2]
    assert _rel(dx.permute(0, 3, 1, 2).cpu(), want) < 2e-5


def test_gconv_split_fused_epilogue():
    """bias + addend + activation on the first act_cols channels (the inference form, rd_gconv_fused's contract)."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 64, 96, 21, 30
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * 0.05
    bias = torch.randn(co, generator=g)
    add = torch.randn(n, co, h, w, generator=g)
    y = F.conv2d(x, wt, padding=1) + bias[None, :, None, None] + add
    y[:, :64] = F.relu(y[:, :64])
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    out = torch.full((n, h, w, co), float("nan"), device="cuda")
    ops.gconv_split(d, ops.nchw_to_nhwc(x.cuda()), ops.pack_weights_split(wt.cuda()), out, bias=bias.cuda(), act=1, act_cols=64,
                    addend=ops.nchw_to_nhwc(add.cuda()), ld_add=co)
    torch.cuda.synchronize()
    assert _rel(out.permute(0, 3, 1, 2).cpu(), y) < 2e-5


@pytest.mark.parametrize("cfg", [(2, 64, 64, 3, 113, 200), (2, 512, 512, 3, 15, 25), (4, 640, 512, 1, 15, 25), (2, 256, 256, 3, 29, 50)])
def test_split_is_as_close_to_fp64_as_the_fp32_mfma(cfg):
    """Error against an fp64 convolution of the same fp32 inputs: the six-term bf16 reconstruction must not be further from it
    than 1.5x the fp32-MFMA kernel (it rounds the accumulator less often, so it is usually closer), max and RMS."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, h, w = cfg
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y64 = F.conv2d(x.double(), wt.double(), padding=k // 2)
    d = cd.conv_fwd(n, h, w, ci, co, k, 1, k // 2)
    xg = ops.nchw_to_nhwc(x.cuda())
    o32 = torch.empty(n, h, w, co, device="cuda")
    osp = torch.empty(n, h, w, co, device="cuda")
    ops.gconv(d, xg, ops.pack_weights(wt.cuda()), o32)
    ops.gconv_split(d, xg, ops.pack_weights_split(wt.cuda()), osp)
    torch.cuda.synchronize()
    e32 = o32.permute(0, 3, 1, 2).cpu().double() - y64
    esp = osp.permute(0, 3, 1, 2).cpu().double() - y64
    print("fp32 MFMA: max %.3e rms %.3e | split: max %.3e rms %.3e" % (e32.abs().max(), e32.pow(2).mean().sqrt(), esp.abs().max(), esp.pow(2).mean().sqrt()))
    assert esp.abs().max() <= 1.5 * e32.abs().max()
    assert esp.pow(2).mean().sqrt() <= 1.5 * e32.pow(2).mean().sqrt()


def _build(h, w):
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import procedural_fill_
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    return m.cuda()


def test_split_step_matches_oracle():
    """HipTrainStep(operands="split") against the CPU oracle for 3 SGD steps, at the tolerances of the fp32 plan's own test
    (tests/test_gpu_model.py::test_fused_step_matches_oracle): loss 2e-3, parameter norms 5e-3, head weight 5e-3 -- and the first
    step's prediction within the north-star 1e-3 of the oracle's forward map."""
    import numpy as np
    from oracle.criteria import MaskedL1Loss as OL1
    from oracle.models import ResNet_latefusion as ORef
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    b, h, w = 2, 97, 161
    m = _build(h, w)
    torch.manual_seed(0)
    o = ORef(18, "upproj", [h, w], 4, False)
    procedural_fill_(o)
    o.train()
    opt = torch.optim.SGD(o.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    ts = HipTrainStep(m, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, operands="split")
    kinds = [k for k, _ in ts.plan.meta.values()]
    assert kinds.count("gconv_split") + kinds.count("gconv_split_pre") > 40, "the split plan must route its convolutions to rd_gconv_split"
    crit = OL1()
    for it in range(3):
        x, t = make_batch(b, h, w, 99 + it, ref_pixels=h * w)
        po_ = o(x)
        lo = crit(po_, t)
        opt.zero_grad()
        lo.backward()
        opt.step()
        lg, pred = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        if it == 0:
            assert _rel(pred.cpu().reshape(po_.shape), po_.detach()) < 1e-3
        assert abs(lg.item() - lo.item()) / lo.item() < 2e-3, (it, lg.item(), lo.item())
    po = np.array([p.double().norm().item() for p in o.parameters()])
    pg = np.array([p.double().norm().item() for p in m.parameters()])
    assert np.abs(po - pg).max() / po.max() < 5e-3
    a_, c_ = m.conv3.weight.detach().cpu().double(), o.conv3.weight.detach().double()
    assert ((a_ - c_).norm() / c_.norm()).item() < 5e-3


def test_split_step_tracks_the_fp32_step():
    """One step from identical parameters on identical data: the split plan's prediction and loss against the fp32 plan's
    (both are fp32-accurate evaluations of the same network: they differ by rounding only)."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 129, 193
    x, t = make_batch(b, h, w, 5, ref_pixels=h * w)
    res = []
    for operands in ("fp32", "split"):
        m = _build(h, w)
        ts = HipTrainStep(m, b, h, w, operands=operands)
        lg, pred = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        res.append((lg.item(), pred.float().cpu().clone(), torch.cat([p.grad.flatten() if p.grad is not None else p.detach().flatten()
                                                                      for p in m.parameters()]).double().norm().item()))
    assert abs(res[0][0] - res[1][0]) / abs(res[0][0]) < 1e-5
    assert _rel(res[1][1], res[0][1]) < 1e-4
    assert abs(res[0][2] - res[1][2]) / res[0][2] < 1e-4


def test_split_step_is_bitwise_reproducible():
    """Two independent split plans stepping the same data concurrently on their own streams end with bit-identical parameters and losses
    (every split kernel is deterministic: fixed reduction orders, no atomics)."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m1, m2 = _build(h, w), _build(h, w)
    t1, t2 = HipTrainStep(m1, b, h, w, operands="split"), HipTrainStep(m2, b, h, w, operands="split")
    losses = []
    for it in range(3):
        x, t = make_batch(b, h, w, 7 + it, ref_pixels=h * w)
        l1, _ = t1.step(x.cuda(), t.cuda())
        l2, _ = t2.step(x.cuda(), t.cuda())
        losses.append((l1, l2))
    torch.cuda.synchronize()
    for l1, l2 in losses:
        assert l1.item() == l2.item()
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(p1, p2)


@pytest.mark.slow
def test_multistage_split_step_matches_oracle():
    """HipTrainStep(operands="split") on resnet18_multistage_uncertainty_fixs (two stages, uncertainty-weighted losses, SGD incl. w_stage1/2)
    against the CPU oracle at the fp32 plan's tolerances (tests/test_gpu_model.py::test_multistage_fused_step_matches_oracle)."""
    import types

    import numpy as np
    from oracle import train as otrain
    from radar_depth_amd import main as hmain
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    b, h, w = 2, 97, 161
    args = types.SimpleNamespace(arch="resnet18_multistage_uncertainty_fixs", decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    hm, hw_ = hmain.create_model(args, [h, w])
    om, ow = otrain.create_model(args, [h, w])
    procedural_fill_(hm)
    procedural_fill_(om)
    hm = hm.cuda()
    om.train()
    opt = torch.optim.SGD(om.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit = otrain.make_criterion(args.arch)
    ts = hmain.HipTrainStep(hm, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=hw_, operands="split")
    kinds = [k for pl in ts.plans for k, _ in pl.meta.values()]
    assert kinds.count("gconv_split") + kinds.count("gconv_split_pre") > 80 and kinds.count("wgrad_split") > 40
    for it in range(3):
        x, t = make_batch(b, h, w, 500 + it, ref_pixels=h * w)
        lo, _, _ = otrain.train_step(args.arch, om, crit, opt, x, t, ow)
        lg, _ = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        assert abs(lg.item() - lo.item()) / abs(lo.item()) < 2e-3, (it, lg.item(), lo.item())
    assert abs(hm.w_stage1.item() - om.w_stage1.item()) < 1e-3 and abs(hm.w_stage2.item() - om.w_stage2.item()) < 1e-3
    po = np.array([p.double().norm().item() for p in om.parameters()])
    pg = np.array([p.double().norm().item() for p in hm.parameters()])
    assert np.abs(po - pg).max() / po.max() < 5e-3


def test_split_forward_at_the_bench_geometry():
    """b = 16, 450 x 800 (BASELINE config 2's own size): the split plan's first training-mode prediction and loss against the fp32 plan's
    from identical parameters -- both within the north-star 1e-3 of each other by a wide margin."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 16, 450, 800
    x, t = make_batch(b, h, w, 1234)
    res = []
    for operands in ("fp32", "split"):
        m = _build(h, w)
        ts = HipTrainStep(m, b, h, w, operands=operands)
        lg, pred = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        res.append((lg.item(), pred.float().cpu().clone()))
        ts.close()
        del ts, m
        torch.cuda.empty_cache()
    assert abs(res[0][0] - res[1][0]) / abs(res[0][0]) < 1e-5
    assert _rel(res[1][1], res[0][1]) < 1e-4


@pytest.mark.parametrize("ex,ew", [(60, -60), (-60, 60), (-30, -30), (40, 0), (-100, 100)])
def test_split_dynamic_range(ex, ew):
    """Operands scaled by 2^+-60 (and the activation down to 2^-100: its third piece 2^-16 |x| is still a normal bf16 number): the
    split convolution against an fp64 convolution of the same fp32 operands, at the kernel's own 2e-5 bar, side by side with rd_gconv."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 128, 128, 29, 50
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g) * 2.0 ** ex
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5 * 2.0 ** ew
    y64 = F.conv2d(x.double(), wt.double(), padding=1)
    assert torch.isfinite(y64.float()).all()
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    xg = ops.nchw_to_nhwc(x.cuda())
    osp = torch.empty(n, h, w, co, device="cuda")
    o32 = torch.empty(n, h, w, co, device="cuda")
    ops.gconv_split(d, xg, ops.pack_weights_split(wt.cuda()), osp)
    ops.gconv(d, xg, ops.pack_weights(wt.cuda()), o32)
    torch.cuda.synchronize()
    esp = _rel(osp.permute(0, 3, 1, 2).cpu().double(), y64)
    e32 = _rel(o32.permute(0, 3, 1, 2).cpu().double(), y64)
    print("scale 2^%d x 2^%d: split %.3e, fp32 MFMA %.3e" % (ex, ew, esp, e32))
    assert esp < 2e-5 and esp <= 1.5 * e32 + 1e-7


def test_split_dynamic_range_and_non_finite():
    """Non-finite activations (include/radar_depth_hip.h, rd_gconv_split): an output that touches a NaN / +-inf input is non-finite
    in both kernels (inf - bf16(inf) = NaN, so the split kernel reports NaN where rd_gconv reports +-inf); every output outside the
    3x3 footprint of the poisoned pixels is bit-for-bit what the clean input gives."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 64, 64, 40, 60
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
    bad = [(0, 5, 10, 20, float("inf")), (1, 33, 0, 0, float("-inf")), (1, 7, 39, 59, float("nan"))]
    xb = x.clone()
    touched = torch.zeros(n, h, w, dtype=torch.bool)
    for (i, c, r, q, v) in bad:
        xb[i, c, r, q] = v
        touched[i, max(r - 1, 0):r + 2, max(q - 1, 0):q + 2] = True
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    wp, wp32 = ops.pack_weights_split(wt.cuda()), ops.pack_weights(wt.cuda())
    outs = []
    for src in (x, xb):
        osp = torch.empty(n, h, w, co, device="cuda")
        o32 = torch.empty(n, h, w, co, device="cuda")
        ops.gconv_split(d, ops.nchw_to_nhwc(src.cuda()), wp, osp)
        ops.gconv(d, ops.nchw_to_nhwc(src.cuda()), wp32, o32)
        torch.cuda.synchronize()
        outs.append((osp.cpu(), o32.cpu()))
    (clean_sp, _), (bad_sp, bad_32) = outs
    assert torch.isfinite(clean_sp).all()
    assert not torch.isfinite(bad_sp[touched]).any() and not torch.isfinite(bad_32[touched]).any()
    assert torch.isnan(bad_sp[touched]).all()                       # (the fp32 MFMA keeps +-inf where no NaN joins the sum)
    assert torch.equal(bad_sp[~touched], clean_sp[~touched])


# ------------------------------------------------------------------------------------------------ pre-split activations
def test_split_pieces_are_exact():
    """rd_split_pieces: three bf16 planes [piece][C/16][M][16] whose sum is the fp32 value EXACTLY, for values of every magnitude."""
    from radar_depth_amd import ops
    g = torch.Generator().manual_seed(21)
    x = torch.randn(3, 7, 11, 48, generator=g) * torch.exp2(torch.randint(-40, 40, (3, 7, 11, 48), generator=g).float())
    x[0, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.5])
    pc = ops.split_pieces(x.cuda())
    torch.cuda.synchronize()
    m = 3 * 7 * 11
    back = pc.double().sum(0).permute(1, 0, 2).reshape(m, 48)          # [C/16][M][16] -> [M][C]
    assert torch.equal(back.float().cpu(), x.reshape(m, 48))
    assert torch.equal(back.cpu(), x.reshape(m, 48).double())           # the three pieces add up without any rounding
    p0 = pc[0].permute(1, 0, 2).reshape(m, 48).float().cpu()
    assert torch.equal(p0, x.reshape(m, 48).to(torch.bfloat16).float())  # piece 0 is the round-to-nearest-even bf16 value


PRE_FWD = [
    (2, 64, 64, 3, 1, 1, 113, 200),   # layer1
    (2, 128, 128, 3, 1, 1, 57, 100),
    (2, 256, 256, 3, 1, 1, 29, 50),
    (2, 512, 512, 3, 1, 1, 15, 25),   # layer4
    (2, 64, 128, 3, 2, 1, 113, 200),  # stride 2: wide patch
    (2, 128, 256, 3, 2, 1, 57, 100),  # layer3.0.conv1 / layer4.0.conv1: the stride-2 layers the default plan routes here
    (16, 256, 512, 3, 2, 1, 29, 50),
    (2, 640, 512, 1, 1, 0, 15, 25),   # conv_fusion
    (2, 32, 32, 3, 1, 1, 120, 200),   # decoder.layer3 conv2
    (3, 32, 48, 3, 1, 1, 9, 7),       # tiny / ragged
    (2, 48, 80, 3, 1, 1, 31, 17),     # channels that do not fill a block, 48-channel reduction
    (1, 96, 48, 3, 2, 1, 33, 45),
    (3, 64, 64, 3, 1, 1, 1, 1),
    (2, 32, 32, 3, 1, 1, 40, 1),
    (16, 64, 64, 3, 1, 1, 57, 100),
]


@pytest.mark.parametrize("cfg", PRE_FWD)
def test_gconv_split_pre_forward(cfg):
    """rd_gconv_split_pre (activation split by its producer, staging = global_load_lds only) at rd_gconv_split's own bar, statistics
    partials included; NaN-filled output must be fully overwritten."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
    if not ops.gconv_split_pre_supported(d):
        pytest.skip("no pre-split plan for this descriptor")
    xp = ops.split_pieces(ops.nchw_to_nhwc(x.cuda()))
    wp = ops.pack_weights_split(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
    stat = torch.zeros(ops.gconv_split_pre_stat_tiles(d), 2, co, device="cuda")
    ops.gconv_split_pre(d, xp, wp, out, stat=stat)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, y) < 2e-5, (_rel(got, y), cfg)
    s_ = stat.sum(0).cpu().double()
    ref_q = (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - y.double().sum((0, 2, 3))).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert _rel(s_[1], ref_q) < 1e-4


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 3, 1, 1, 113, 200),
    (2, 64, 128, 3, 2, 1, 113, 200),
    (2, 256, 512, 3, 2, 1, 29, 50),
    (16, 512, 512, 3, 1, 1, 15, 25),
    (2, 48, 80, 3, 1, 1, 31, 17),
    (3, 64, 64, 3, 1, 1, 1, 1),
])
def test_gconv_split_pre_dgrad(cfg):
    """Input gradient through the pre-split form (dy split by its producer), with a residual-gradient addend in the epilogue."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
    if not ops.gconv_split_pre_supported(dd):
        pytest.skip("no pre-split plan for this descriptor")
    add = torch.randn(n, h, w, ci, generator=g) if not zero_fill else None
    dx = torch.zeros(n, h, w, ci, device="cuda") if zero_fill else torch.full((n, h, w, ci), float("nan"), device="cuda")
    gp = ops.split_pieces(ops.nchw_to_nhwc(gy.cuda()))
    addg = add.cuda() if add is not None else None
    ops.gconv_split_pre(dd, gp, ops.pack_weights_split(wt.cuda(), transpose=True), dx, addend=addg, ld_add=ci if add is not None else 0)
    torch.cuda.synchronize()
    want = x.grad + (add.permute(0, 3, 1, 2) if add is not None else 0)
    got = dx.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, want) < 2e-5, (_rel(got, want), cfg)


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (32, 13, 9)])
def test_gconv_split_pre_upproj(c, h, w):
    """UpProj 4-phase forward (9/6/6/4 taps on the low-resolution input) and its 25-tap stride-2 input gradient, pre-split form."""
    from radar_depth_amd import convdesc as cd, ops
    n = 2
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = torch.randn(c, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    u = torch.zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x
    y = F.conv2d(u, wt, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, dd = cd.upproj_fwd(n, h, w, c, c), cd.upproj_dgrad(n, h, w, c, c)
    if not (ops.gconv_split_pre_supported(d) and ops.gconv_split_pre_supported(dd)):
        pytest.skip("no pre-split plan")
    out = torch.full((n, 2 * h, 2 * w, c), float("nan"), device="cuda")
    ops.gconv_split_pre(d, ops.split_pieces(ops.nchw_to_nhwc(x.detach().cuda())), ops.pack_weights_split(wt.cuda()), out)
    dx = torch.full((n, h, w, c), float("nan"), device="cuda")
    ops.gconv_split_pre(dd, ops.split_pieces(ops.nchw_to_nhwc(gy.cuda())), ops.pack_weights_split(wt.cuda(), transpose=True), dx)
    torch.cuda.synchronize()
    assert _rel(out.permute(0, 3, 1, 2).cpu(), y.detach()) < 2e-5
    assert _rel(dx.permute(0, 3, 1, 2).cpu(), x.grad) < 2e-5


def test_gconv_split_pre_launches_are_bitwise_reproducible():
    """200 launches of two shapes under NaN-poisoned LDS: every output bit-identical to the first (the copies' completion is the
    only thing that orders the staging against the matrix waves: a missing wait shows up here)."""
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import lib
    g = torch.Generator().manual_seed(5)
    for (n, ci, co, h, w) in ((2, 64, 64, 57, 100), (2, 256, 256, 29, 50)):
        x = torch.randn(n, h, w, ci, generator=g).cuda()
        wp = ops.pack_weights_split((torch.randn(co, ci, 3, 3, generator=g) * 0.05).cuda())
        d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
        xp = ops.split_pieces(x)
        first = None
        for it in range(100):
            lib().rd_debug_poison_lds(ops.current_stream())
            out = torch.empty(n, h, w, co, device="cuda")
            ops.gconv_split_pre(d, xp, wp, out)
            if first is None:
                first = out.clone()
            else:
                assert torch.equal(out, first), it
    torch.cuda.synchronize()
