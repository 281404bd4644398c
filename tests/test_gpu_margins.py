"""End-to-end parity with the margins on file: every test here PRINTS its worst / median errors; tools/collect_round.sh
keeps them as profiles/<round>_parity_margins.txt (pytest -s).

  * fp64-anchored gradient bar: per parameter tensor, || g_HIP - g_fp64 || against || g_oracle32 - g_fp64 || (the CPU oracle evaluated in
    double on the same inputs) -- the end-to-end counterpart of the kernel-level "as close to fp64 as the fp32 kernel" tests; it takes
    the ReLU-flip conditioning out of the comparison, because the fp32 oracle pays it too;
  * three SGD steps at BASELINE configs[1]'s own size (b = 16, 450 x 800) against the oracle's three steps: state that lives across steps
    at full size (plan.persistent zero buffers, the stream-side weight pack, piece planes) is exercised where the metric is quoted;
  * NaN-filled plan buffers (tools/poison_global.py as a test): every buffer is written before it is read.
"""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(h, w):
    from oracle.models import ResNet_latefusion as ORef
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import procedural_fill_
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    o = ORef(18, "upproj", [h, w], 4, False)
    procedural_fill_(o)
    return m.cuda().train(), o.train()


@pytest.mark.parametrize("operands", ["fp32", "split"])
def test_gradients_as_close_to_fp64_as_the_fp32_oracle(operands):
    """b = 2, 97 x 161, one backward pass.  g64: the oracle in double; g32: the oracle in fp32 (the reference's arithmetic); g: the fused
    step's gradient arena.  What the table (tools/diag_fp64.py, profiles/r05_diag_fp64.txt) shows: every arithmetic -- the CPU fp32
    oracle, the fp32-MFMA plan, the split plan -- is EITHER at ~1e-5 of the fp64 gradient in every tensor (no discrete decision of the
    network differs from the fp64 run: the fp32 plan at this geometry) OR at ~1e-2 in the deep tensors and 1e-4 at the top of the
    decoder (one sign(pred - target) / ReLU decision differs and the chaotic network amplifies it: the CPU oracle and the split plan at
    this geometry, all three at 129 x 193), with the head weight at 2e-7 in all of them.  Which case a run lands in is a property of
    geometry and seed, not of the plan.  The bar is therefore the fp32 oracle's own distance: over all tensors together the HIP
    gradient may be at most 4x as far from fp64 as the fp32 oracle is, per tensor 4x + 1e-2 of the tensor's norm, and the head
    weight (no decision behind it) within 1e-5."""
    import copy
    from oracle.criteria import MaskedL1Loss as OL1
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m, o32 = _pair(h, w)
    o64 = copy.deepcopy(o32).double()
    x, t = make_batch(b, h, w, 99, ref_pixels=h * w)
    OL1()(o32(x), t).backward()
    OL1()(o64(x.double()), t.double()).backward()
    ts = HipTrainStep(m, b, h, w, operands=operands)
    ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    names = [n for n, _ in m.named_parameters()]
    g = [m._grad_view(p).detach().cpu().double() for p in m.parameters()]
    g32 = [p.grad.double() for p in o32.parameters()]
    g64 = [p.grad for p in o64.parameters()]
    n64 = np.array([c.norm().item() for c in g64])
    keep = n64 > 1e-9 * n64.max()           # (bn_fusion.bias: its gradient is zero up to rounding -- conv2 follows without activation)
    e_hip = np.array([(a - c).norm().item() for a, c in zip(g, g64)])[keep]
    e_o32 = np.array([(a - c).norm().item() for a, c in zip(g32, g64)])[keep]
    names = [n for n, k in zip(names, keep) if k]
    n64 = n64[keep]
    ratio = e_hip / (4.0 * e_o32 + 1e-2 * n64)
    k = int(ratio.argmax())
    agg = np.sqrt((e_hip ** 2).sum()) / np.sqrt((e_o32 ** 2).sum())
    print("fp64-anchored gradients [%s]: || g - g64 || over all tensors HIP / oracle32 = %.3g; per tensor rel-to-|g64|: HIP worst %.3e median %.3e, oracle32 worst %.3e "
          "median %.3e; tightest tensor %s at %.2f of its bar; head weight %.2e"
          % (operands, agg, (e_hip / n64).max(), np.median(e_hip / n64), (e_o32 / n64).max(), np.median(e_o32 / n64), names[k], ratio[k],
             e_hip[names.index("conv3.weight")] / n64[names.index("conv3.weight")]))
    bad = [(n, a, c) for n, a, c, r in zip(names, e_hip, e_o32, ratio) if r > 1.0]
    assert not bad, bad[:8]
    assert agg <= 4.0
    assert e_hip[names.index("conv3.weight")] <= 1e-5 * n64[names.index("conv3.weight")]


@pytest.mark.parametrize("b,h,w", [(2, 97, 161), (1, 65, 97)])
def test_split_plan_gradients_below_the_decision_floor_over_seeds(b, h, w):
    """the single-seed test above cannot tell one flipped ReLU / sign(pred - target) decision (amplified by the network
    to ~1e-2 in the deep tensors) from a genuine 1e-3-level defect of a deep layer's gradient in the split plan.  Eight seeds each at
    b = 2, 97 x 161 (where nearly every run of every arithmetic flips something: ~10^7 activations at relative distances of 1e-6) and
    at b = 1, 65 x 97 (fewer activations, more decision-free runs); per seed the worst per-tensor distance to the fp64 oracle's
    gradient, relative to the tensor's norm, for the CPU fp32 oracle (the reference's arithmetic), the fp32-MFMA plan and the split plan.
    An arithmetic is CLEAN on a seed when it is within 1e-4 in every tensor (no discrete decision differs from the fp64 run).  Asserted:
      * a clean run of the split plan is, in EVERY tensor, within 5e-5 (measured ~1e-5 at 97 x 161) or 3x the clean CPU oracle's own
        distance or 10x the clean fp32-MFMA plan's: below the decision floor nothing of the 1e-3 class exists in any layer's gradient;
      * the split plan is clean at least as often as the fp32-MFMA plan minus one, and as the CPU oracle minus one (decisions flip at
        the same rate as in the other fp32 arithmetics: no extra error pushes activations across their thresholds);
      * on flipped seeds: 4x the oracle's own distance + 3e-2 of the norm (the oracle may be clean where the split plan flipped: a
        single flip measured 2.2e-2 in one tensor at the small geometry)."""
    import copy
    from oracle.criteria import MaskedL1Loss as OL1
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    m, o32 = _pair(h, w)
    o64 = copy.deepcopy(o32).double()
    steps = {op: HipTrainStep(m, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0, operands=op) for op in ("fp32", "split")}   # lr 0: parameters stay put
    names = [n for n, _ in m.named_parameters()]
    rows, clean = [], {"oracle32": 0, "fp32": 0, "split": 0}
    for seed in range(101, 109):
        x, t = make_batch(b, h, w, seed, ref_pixels=h * w)
        for o in (o32, o64):
            o.zero_grad()
        OL1()(o32(x), t).backward()
        OL1()(o64(x.double()), t.double()).backward()
        g64 = [p.grad.clone() for p in o64.parameters()]
        n64 = np.array([c.norm().item() for c in g64])
        keep = n64 > 1e-9 * n64.max()
        dist = {"oracle32": np.array([(p.grad.double() - c).norm().item() for p, c in zip(o32.parameters(), g64)])[keep] / n64[keep]}
        for op, ts in steps.items():
            ts.step(x.cuda(), t.cuda())
            torch.cuda.synchronize()
            g = [m._grad_view(p).detach().cpu().double() for p in m.parameters()]
            dist[op] = np.array([(a - c).norm().item() for a, c in zip(g, g64)])[keep] / n64[keep]
        is_clean = {k: bool(v.max() <= 1e-4) for k, v in dist.items()}
        for k in clean:
            clean[k] += is_clean[k]
        rows.append((seed, dist, is_clean))
        print("[b=%d %dx%d] " % (b, h, w), end="")
        print("seed %d: worst / median per-tensor distance to fp64: oracle32 %.2e / %.2e %s | fp32-MFMA %.2e / %.2e %s | split %.2e / %.2e %s"
              % (seed, dist["oracle32"].max(), np.median(dist["oracle32"]), "clean" if is_clean["oracle32"] else "FLIP ",
                 dist["fp32"].max(), np.median(dist["fp32"]), "clean" if is_clean["fp32"] else "FLIP ",
                 dist["split"].max(), np.median(dist["split"]), "clean" if is_clean["split"] else "FLIP "))
        kept = [n for n, k in zip(names, keep) if k]
        if is_clean["split"]:
            bar = np.full_like(dist["split"], 5e-5)
            if is_clean["fp32"]:
                bar = np.maximum(bar, 10.0 * dist["fp32"])
            if is_clean["oracle32"]:
                bar = np.maximum(bar, 3.0 * dist["oracle32"])
            bad = [(n, a, c) for n, a, c, lim in zip(kept, dist["split"], dist["fp32"], bar) if a > lim]
            assert not bad, (seed, bad[:6])
        else:
            lim = 4.0 * dist["oracle32"] + 3e-2
            bad = [(n, a, c) for n, a, c, l_ in zip(kept, dist["split"], dist["oracle32"], lim) if a > l_]
            assert not bad, (seed, bad[:6])
    print("[b=%d %dx%d] clean seeds of 8 (every tensor within 1e-4 of fp64): CPU oracle fp32 %d, fp32-MFMA plan %d, split plan %d"
          % (b, h, w, clean["oracle32"], clean["fp32"], clean["split"]))
    assert clean["split"] >= clean["fp32"] - 1 and clean["split"] >= clean["oracle32"] - 1, clean
    assert clean["split"] >= 1, "no seed on which the split plan is decision-free at this geometry: pick other seeds"


@pytest.mark.slow
@pytest.mark.parametrize("operands", ["split"])
def test_config2_three_steps_b16_450x800_vs_oracle(operands):
    """BASELINE configs[1]'s size, THREE consecutive SGD steps (lr 0.01, momentum 0.9, weight decay 1e-4: main.py:285-290, 400-447) against
    the CPU oracle's three steps, at the small-geometry bars of tests/test_gpu_gconv_split.py::test_split_step_matches_oracle: loss 2e-3
    per step, parameter norms 5e-3, head weight 5e-3."""
    from oracle.criteria import MaskedL1Loss as OL1
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 16, 450, 800
    m, o = _pair(h, w)
    opt = torch.optim.SGD(o.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    ts = HipTrainStep(m, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, operands=operands)
    crit = OL1()
    errs = []
    for it in range(3):
        x, t = make_batch(b, h, w, 4321 + it)
        lo = crit(o(x), t)
        opt.zero_grad()
        lo.backward()
        opt.step()
        lg, _ = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        errs.append(abs(lg.item() - lo.item()) / lo.item())
        assert errs[-1] < 2e-3, (it, lg.item(), lo.item())
    po = np.array([p.double().norm().item() for p in o.parameters()])
    pg = np.array([p.double().norm().item() for p in m.parameters()])
    a_, c_ = m.conv3.weight.detach().cpu().double(), o.conv3.weight.detach().double()
    head = ((a_ - c_).norm() / c_.norm()).item()
    print("config2 b=16 450x800 three steps [%s]: loss rel err per step %s, parameter norms worst %.3e (of the largest), head weight %.3e"
          % (operands, " ".join("%.2e" % e for e in errs), np.abs(po - pg).max() / po.max(), head))
    assert np.abs(po - pg).max() / po.max() < 5e-3
    assert head < 5e-3


def _build(arch, h, w):
    from radar_depth_amd.main import create_model
    from radar_depth_amd.synthetic import procedural_fill_
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    made = create_model(args, [h, w])
    m, lw = made if isinstance(made, tuple) else (made, None)
    procedural_fill_(m)
    return m.cuda(), lw


@pytest.mark.parametrize("arch,b,h,w", [("resnet18_latefusion", 5, 97, 161), ("resnet18_latefusion", 2, 450, 800),
                                        ("resnet18_multistage_uncertainty_fixs", 2, 129, 193)])
def test_nan_filled_plan_buffers_do_not_change_the_step(arch, b, h, w):
    """Every fp32 buffer of the plan except its input and its build-time state (plan.persistent: the zeros between a strided input
    gradient's pixels) is NaN-filled before the first step; two steps must still give the loss and parameters of a clean instance bit
    for bit, and the persistent zero buffers must still hold zeros wherever the plan does not write (a stray write there
    would corrupt every later step silently)."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    (m1, lw1), (m2, lw2) = _build(arch, h, w), _build(arch, h, w)
    t1 = HipTrainStep(m1, b, h, w, loss_weights=lw1)
    t2 = HipTrainStep(m2, b, h, w, loss_weights=lw2)
    n = 0
    for plan in t2.plans:
        state = {plan.x_in.data_ptr()} | {p.data_ptr() for p in getattr(plan, "persistent", [])}
        for t in plan.keep:
            if torch.is_tensor(t) and t.dtype == torch.float32 and t.data_ptr() not in state:
                t.fill_(float("nan"))
                n += 1
    assert n > 50
    for it in range(2):
        x, t = make_batch(b, h, w, 70 + it, ref_pixels=h * w)
        l1, _ = t1.step(x.cuda(), t.cuda())
        l2, _ = t2.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        assert l1.item() == l2.item(), (it, l1.item(), l2.item())
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(p, q)
    for plan in t2.plans:
        for chk in getattr(plan, "persistent_zero_checks", []):
            assert chk(), "a persistent zero buffer was written where the plan must never write"
