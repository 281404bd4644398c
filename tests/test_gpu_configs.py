"""Parity on BASELINE.json's own configurations (the sizes the metric is quoted on), plus layer-level forward/backward pins.

  config 2  resnet18_latefusion b=16 450x800 fp32          forward + loss vs the live CPU oracle          (<= 1e-3 / 1e-4)
  config 4  multistage_uncertainty_fixs 450x800            b=2 vs the reference-generated golden fixture   (multistage_full.npz)
                                                           b=8 forward + losses vs the live CPU oracle
  config 5  multistage bf16, 900x1600                      bf16 multistage training step vs the oracle with the same rounding
                                                           points (97x161); fp32 multistage 900x1600 b=1 vs the oracle;
                                                           bf16 multistage 900x1600 b=1 step vs the emulated oracle's losses
  layers    one UpProjModule / BasicBlock fwd+bwd through the plan's builders vs upproj_module.npz / basic_block.npz (1e-4)
  misc      plain resnet18_multistage (loss = d1 + d2, main.py:431-438); MaskedMSELoss (`-c l2`)
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _t(x):
    return x.detach().cpu().numpy()


def _latefusion_pair(h, w):
    from oracle.models import ResNet_latefusion as ORef
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import procedural_fill_
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    o = ORef(18, "upproj", [h, w], 4, False)
    procedural_fill_(o)
    return m.cuda().train(), o.train()


def _multistage_pair(h, w, arch="resnet18_multistage_uncertainty_fixs"):
    from oracle import train as otrain
    from radar_depth_amd import main as hmain
    from radar_depth_amd.synthetic import procedural_fill_
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    made_h, made_o = hmain.create_model(args, [h, w]), otrain.create_model(args, [h, w])
    hm, hw_ = made_h if isinstance(made_h, tuple) else (made_h, None)
    om, ow = made_o if isinstance(made_o, tuple) else (made_o, None)
    procedural_fill_(hm)
    procedural_fill_(om)
    return args, hm.cuda().train(), hw_, om.train(), ow


_ORACLE_CACHE = {}


def _oracle_config2():
    """CPU oracle at BASELINE configs[1]'s own size (b = 16, 450 x 800, seed 1234): training-mode forward, MaskedL1, backward -- run ONCE per
    session and shared by the two plans' tests (the suite's run time is dominated by repeated CPU oracle steps)."""
    if "c2" not in _ORACLE_CACHE:
        from oracle.criteria import MaskedL1Loss as OL1
        from radar_depth_amd.synthetic import make_batch
        _, o = _latefusion_pair(450, 800)
        x, t = make_batch(16, 450, 800, 1234)
        yo = o(x)
        lo = OL1()(yo, t)
        sd = {k: v.clone() for k, v in o.state_dict().items() if "running" in k}
        lo.backward()
        _ORACLE_CACHE["c2"] = dict(yo=yo.detach(), lo=lo.item(), sd=sd, names=[n for n, _ in o.named_parameters()],
                                   gn=np.array([p.grad.double().norm().item() for p in o.parameters()]), head=_t(o.conv3.weight.grad))
    return _ORACLE_CACHE["c2"]


# ------------------------------------------------------------------------------------------------ config 2
@pytest.mark.parametrize("operands", ["fp32", "split"])
def test_config2_latefusion_b16_450x800_vs_oracle(operands):
    """BASELINE configs[1] exactly: b=16, 450x800, fp32, train mode (batch statistics over 16 samples).  Forward map within
    1e-3 of the CPU oracle's (north_star), loss within 1e-4, then one fused step must leave the same loss in the step's
    own loss slot and a finite, changed parameter set.  Both plans of the fused step at the SAME bars: "fp32" (every convolution
    on the fp32 MFMA) and "split" (the default: fp32 operands as three bf16 pieces on the bf16 matrix cores)."""
    from oracle.criteria import MaskedL1Loss as OL1
    from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 16, 450, 800
    m, _ = _latefusion_pair(h, w)
    m.operands = operands                       # (the eager forward below runs on the same arithmetic as the fused step)
    x, t = make_batch(b, h, w, 1234)
    orc = _oracle_config2()                     # (the CPU oracle's forward + backward at b=16: ~15 s, computed once for both plans)
    yo, lo, o_sd, o_names, on, oh = orc["yo"], orc["lo"], orc["sd"], orc["names"], orc["gn"], orc["head"]
    with torch.no_grad():
        y = m(x.cuda())
        lg = MaskedL1Loss()(y, t.cuda())
    e = rel(_t(y), _t(yo))
    assert e < 1e-3, e
    assert abs(lg.item() - lo) / lo < 1e-4
    # running statistics of the first and the last BatchNorm after that one training-mode forward
    for k in ("bn1.running_mean", "bn1.running_var", "decoder.layer4.upper_branch.batchnorm2.running_var", "bn_fusion.running_mean"):
        assert rel(_t(m.state_dict()[k]), _t(o_sd[k])) < 1e-3, k
    before = m.conv3.weight.detach().clone()
    ts = HipTrainStep(m, b, h, w, operands=operands)
    if operands == "split":
        kinds = [k for k, _ in ts.plan.meta.values()]
        assert kinds.count("gconv_split") + kinds.count("gconv_split_pre") >= 40 and kinds.count("wgrad_split") >= 20, "the split plan must run on rd_gconv_split / rd_wgrad_split"
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    # the fused step's forward saw BN running stats one update later, which do not enter train-mode outputs: same loss
    assert abs(loss.item() - lo) / lo < 1e-4
    assert rel(_t(pred), _t(yo)) < 1e-3
    assert not torch.equal(before, m.conv3.weight) and all(torch.isfinite(p).all() for p in m.parameters())
    # BACKWARD at config 2's own batch: the gradient arena the fused step left behind (zero_grad + backward, main.py:443-444)
    # against the oracle's autograd at b=16 -- every parameter tensor's gradient norm, and the well-conditioned head gradient
    # element-wise.  2e-2: the same conditioning-limited bound as the b=2 golden check (single ReLU flips at |z| ~ 1e-6 of the
    # map's max move the deepest tensors by about a percent; tools/diag_bwd.py), everything shallow sits below 1e-3.
    names = [n for n, _ in m.named_parameters()]
    assert names == o_names
    gn = np.array([m._grad_view(p).double().norm().item() for p in m.parameters()])
    floor = 1e-6 * on.max()
    bad = [(n, a, c) for n, a, c in zip(names, gn, on) if abs(a - c) > 2e-2 * c + floor]
    print("config2 b=16 [%s] gradient norms: worst rel %.3e, median rel %.3e" % (operands, np.max(np.abs(gn - on) / (on + floor)), np.median(np.abs(gn - on) / (on + floor))))
    assert not bad, bad[:8]
    gh = _t(m._grad_view(m.conv3.weight))
    assert np.abs(gh - oh).max() <= 1e-2 * np.abs(oh).max()


# ------------------------------------------------------------------------------------------------ config 4
@pytest.mark.parametrize("operands", ["split", "fp32"])
def test_config4_multistage_450x800_vs_golden(operands):
    """multistage_uncertainty_fixs at 450x800 (b=2) against vectors generated from the real reference
    (model/multistage_model.py:63-83, main.py:416-429): the four output maps (strided by 8), the filter mask, the three loss
    terms + total, d(total)/d(w_stage1,2), every parameter's gradient norm, two full gradients, parameter norms after SGD.
    Through the drop-in surface (`o = model(x)`; torch autograd; torch.optim.SGD) on both eager plans (model.operands)."""
    from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss, SmoothnessLoss
    from radar_depth_amd.model.multistage_model import ResNet_multistage
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    want = np.load(os.path.join(GOLD, "multistage_full.npz"))
    b, h, w, sub = 2, 450, 800, 8
    torch.manual_seed(0)
    m = ResNet_multistage(18, "upproj", [h, w], False)
    w1, w2 = torch.nn.Parameter(torch.tensor(1.0)), torch.nn.Parameter(torch.tensor(1.0))
    m.register_parameter("w_stage1", w1)
    m.register_parameter("w_stage2", w2)
    procedural_fill_(m)
    m = m.cuda().train()
    m.operands = operands
    x, t = make_batch(b, h, w, 4242)
    x[:, 3, ::7, ::11] = torch.where(x[:, 3, ::7, ::11] > 0, x[:, 3, ::7, ::11], torch.full_like(x[:, 3, ::7, ::11], 60.0))
    x, t = x.cuda(), t.cuda()
    assert [n for n, _ in m.named_parameters()] == list(want["param_names"])
    opt = torch.optim.SGD(m.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit, smooth = MaskedL1Loss(), SmoothnessLoss()
    o = m(x)
    for k in ("stage1", "stage2", "radar_filtered"):
        assert rel(_t(o[k])[:, :, ::sub, ::sub], want["out/" + k]) < 1e-3, k
    assert (_t(o["mask"])[:, :, ::sub, ::sub] != want["out/mask"]).mean() < 1e-4
    assert abs(o["mask"].mean().item() - want["mask_density"][0]) < 1e-5
    d1, d2, sm = crit(o["stage1"], t), crit(o["stage2"], t), smooth(o["stage1"], x)
    W1, W2 = m.w_stage1, m.w_stage2
    loss = torch.exp(-W1) * (d1 + 0.1 * sm) + torch.exp(-W2) * d2 + (W1 + W2)
    got = np.array([d1.item(), d2.item(), sm.item(), loss.item()])
    assert np.abs(got - want["losses"]).max() / np.abs(want["losses"]).max() < 1e-4, (got, want["losses"])
    opt.zero_grad()
    loss.backward()
    assert np.abs(np.array([W1.grad.item(), W2.grad.item()]) - want["w_grads"]).max() < 1e-4 * np.abs(want["w_grads"]).max()
    gn = np.array([p.grad.double().norm().item() for p in m.parameters()])
    floor = 1e-6 * want["grad_norms"].max()
    # (3e-2: the deepest tensors -- stage 1's depth stem and its BatchNorm, behind stage 2, the radar filter and all of stage 1 --
    #  move by 1-2 % with the summation order of the kernels in between: 1.6e-2 ... 2.2e-2 over kernel revisions, everything else < 1e-2)
    bad = [(n, a, c) for n, a, c in zip(want["param_names"], gn, want["grad_norms"]) if abs(a - c) > 3e-2 * c + floor]
    assert not bad, bad[:8]
    # element-wise: the head weight (well conditioned) at 3e-2 of its max; stage 2's depth stem sits behind the whole stage-2 depth
    # encoder backward, where single ReLU flips at |z| ~ 1e-6 of the map's max move individual elements by percents
    # (tests/test_gpu_model.py docstring, tools/diag_bwd.py): 6e-2 of the max and 5e-2 norm-wise on the fp32-MFMA plan (measured 3.7e-2 /
    # 3.7e-2 in round 5).  The split plan, the eager default since round 6, measured 6.3e-2 / 4.9e-2 on this one tensor -- a different
    # set of flipped decisions, not a less accurate arithmetic: tests/test_gpu_margins.py::test_split_plan_gradients_below_the_decision_
    # floor_over_seeds pins the split plan to the fp32 plan's accuracy tensor by tensor wherever no decision flips, and counts the flips of
    # both plans over eight seeds.  Its bar here is therefore stated separately (1e-1 / 8e-2), not folded into the fp32 plan's.
    stem_tol = (6e-2, 5e-2) if operands == "fp32" else (1e-1, 8e-2)
    for k, tol, ntol in (("stage1.conv3.weight", 3e-2, 5e-2), ("stage2.conv1_depth.weight",) + stem_tol):
        g = _t(dict(m.named_parameters())[k].grad)
        e_max = np.abs(g - want["grad/" + k]).max() / np.abs(want["grad/" + k]).max()
        e_nrm = np.linalg.norm(g - want["grad/" + k]) / np.linalg.norm(want["grad/" + k])
        print("config4 grad [%s] %s: max-rel %.3e norm-rel %.3e" % (operands, k, e_max, e_nrm))
        assert e_max <= tol and e_nrm <= ntol, (k, e_max, e_nrm)
    opt.step()
    pn = np.array([p.double().norm().item() for p in m.parameters()])
    assert np.abs(pn - want["param_norms1"]).max() / want["param_norms1"].max() < 1e-4


@pytest.mark.parametrize("operands", ["fp32", "split"])
def test_config4_multistage_b8_450x800_fused_step_vs_oracle(operands):
    """BASELINE configs[3] exactly (b=8, 450x800): the fused multistage step's forward maps, its four loss terms and -- BACKWARD at
    the configuration's own batch -- every parameter tensor's gradient norm against the live CPU oracle's training-mode forward and
    autograd (lr = 1, no momentum, no weight decay: the update IS the gradient).  Both plans of the fused step at the same bars."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 8, 450, 800
    args, hm, hw_, _, _ = _multistage_pair(h, w)
    x, t = make_batch(b, h, w, 1234)
    if "c4" not in _ORACLE_CACHE:                 # (the CPU oracle's multistage forward + backward at b=8: ~18 s, once for both plans)
        _, _, _, om, ow = _multistage_pair(h, w)
        crit = otrain.make_criterion(args.arch)
        lo_, po_, ex_ = otrain.compute_loss(args.arch, om, crit, x, t, ow)
        lo_.backward()
        _ORACLE_CACHE["c4"] = dict(lo=lo_.item(), po=po_.detach(), ex={k: v.detach() for k, v in ex_.items() if torch.is_tensor(v)},
                                   names=[n for n, _ in om.named_parameters()],
                                   gn=np.array([p.grad.double().norm().item() for p in om.parameters()]),
                                   head=_t(dict(om.named_parameters())["stage2.conv3.weight"].grad))
    orc = _ORACLE_CACHE["c4"]
    lo, po, ex = orc["lo"], orc["po"], orc["ex"]
    init = [p.detach().clone() for p in hm.parameters()]
    ts = HipTrainStep(hm, b, h, w, lr=1.0, momentum=0.0, weight_decay=0.0, loss_weights=hw_, operands=operands)
    if operands == "split":
        kinds = [k for pl in ts.plans for k, _ in pl.meta.values()]
        assert kinds.count("gconv_split") + kinds.count("gconv_split_pre") >= 80 and kinds.count("wgrad_split") >= 40
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert rel(_t(pred), _t(po)) < 1e-3
    assert rel(_t(ts.mp.p1.pred), _t(ex["pred1"])) < 1e-3
    want4 = np.array([ex["d1"].item(), ex["d2"].item(), ex["smooth"].item(), lo])
    got4 = _t(ts.loss4)
    assert np.abs(got4 - want4).max() / np.abs(want4).max() < 1e-4, (got4, want4)
    names, go = orc["names"], orc["gn"]
    gg = np.array([(i0 - p.detach()).double().norm().item() for i0, p in zip(init, hm.parameters())])
    floor = 1e-6 * go.max()
    worst = np.abs(gg - go) / (go + floor)
    print("config4 b=8 [%s] gradient norms: worst rel %.3e (%s), median rel %.3e" % (operands, worst.max(), names[int(worst.argmax())], np.median(worst)))
    # 3e-2: the bound of the reference-generated b=2 fixture above (stage 1's deepest tensors sit behind stage 2, the radar filter and
    # all of stage 1; single ReLU flips move them by 1-2 %), everything shallow is far below
    bad = [(n, a, c) for n, a, c in zip(names, gg, go) if abs(a - c) > 3e-2 * c + floor]
    assert not bad, bad[:8]
    k3 = names.index("stage2.conv3.weight")
    gh = _t(init[k3] - list(hm.parameters())[k3].detach())
    oh = orc["head"]
    assert np.abs(gh - oh).max() <= 1e-2 * np.abs(oh).max()


# ------------------------------------------------------------------------------------------------ config 5
@pytest.mark.slow
def test_config5_multistage_900x1600_fp32_vs_oracle():
    """configs[4]'s geometry in fp32, b=1: both stages' maps, the loss terms and gradient norms vs the live CPU oracle."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 1, 900, 1600
    args, hm, hw_, om, ow = _multistage_pair(h, w)
    x, t = make_batch(b, h, w, 55)
    crit = otrain.make_criterion(args.arch)
    lo, po, ex = otrain.compute_loss(args.arch, om, crit, x, t, ow)
    lo.backward()
    init = [p.detach().clone() for p in hm.parameters()]
    ts = HipTrainStep(hm, b, h, w, lr=1.0, momentum=0.0, weight_decay=0.0, loss_weights=hw_)   # update == gradient
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert rel(_t(pred), _t(po)) < 1e-3
    assert rel(_t(ts.mp.p1.pred), _t(ex["pred1"])) < 1e-3
    assert abs(loss.item() - lo.item()) / abs(lo.item()) < 1e-4
    go = np.array([p.grad.double().norm().item() for p in om.parameters()])
    gg = np.array([(i0 - p.detach()).double().norm().item() for i0, p in zip(init, hm.parameters())])
    assert np.abs(go - gg).max() / go.max() < 2e-2


def _bf16_emulated(om):
    import importlib.util
    spec = importlib.util.spec_from_file_location("_bf16_tests", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_bf16.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod._emulate_bf16_operands(om)


@pytest.mark.parametrize("geom", [(2, 97, 161), pytest.param((1, 900, 1600), marks=pytest.mark.slow)])
def test_config5_multistage_bf16_step_vs_emulated_oracle(geom):
    """config 5's arithmetic: the multistage_uncertainty_fixs TRAINING STEP with bf16 conv operands against the CPU oracle with
    the same rounding points (tests/test_gpu_bf16.py::_BfConv/_BfStem: forward / input-gradient / >=32-channel weight-gradient
    operands rounded to bf16, fp32 accumulation, everything else fp32).  Stated tolerances: the four loss terms 5e-4; at the small
    geometry additionally the gradient norm of every parameter tensor 6e-2 of the largest and w_stage1/2 gradients 1e-3;
    both maps 0.1 max-norm (pixels that cross a bf16 rounding boundary, see test_gpu_bf16.py) and 4e-2 rms, stage 2
    teacher-forced (see below); the second geometry is config 5's own (900x1600, b=1)."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = geom
    args, hm, hw_, om, ow = _multistage_pair(h, w)
    assert _bf16_emulated(om) == 104
    x, t = make_batch(b, h, w, 600, ref_pixels=h * w if h < 400 else 450 * 800)
    crit = otrain.make_criterion(args.arch)
    lo, po, ex = otrain.compute_loss(args.arch, om, crit, x, t, ow)
    small = h < 400
    if small:
        lo.backward()
    init = [p.detach().clone() for p in hm.parameters()]
    ts = HipTrainStep(hm, b, h, w, lr=1.0, momentum=0.0, weight_decay=0.0, loss_weights=hw_, operands="bf16")
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    want4 = np.array([ex["d1"].item(), ex["d2"].item(), ex["smooth"].item(), lo.item()])
    got4 = _t(ts.loss4)
    e_loss = np.abs(got4 - want4).max() / np.abs(want4).max()
    # Stage 2 is compared TEACHER-FORCED: the oracle's stage 2 (same rounding points) fed with the HIP stage-1 prediction.  End to
    # end, stage 1's bf16 rounding noise (e1, isolated pixels that crossed a bf16 rounding boundary) enters stage 2 through the
    # 7x7 depth stem + a batch-statistics BatchNorm over a nearly constant plane, which amplifies it ~7x (0.21 max / 0.13 rms
    # measured, printed as e2e) -- for ANY bf16 implementation, the fp32 HIP plan included (0.34 against this oracle).
    e1, e2e = rel(_t(ts.mp.p1.pred), _t(ex["pred1"])), rel(_t(pred), _t(po))
    with torch.no_grad():
        p1h = ts.mp.p1.pred.detach().cpu()
        kept_o, mask_o = om.filter_layer(x[:, 3:4], p1h)
        o2 = om.stage2(torch.cat((x[:, :3], kept_o, p1h), 1))
    assert torch.equal(kept_o, ts.mp.kept.cpu()) and torch.equal(mask_o, ts.mp.mask.cpu())     # the radar filter is exact
    e2 = rel(_t(pred), _t(o2))
    r2 = ((pred.detach().cpu() - o2).norm() / o2.norm()).item()
    flips = int((ts.mp.mask.cpu() != ex["out"]["mask"]).sum().item())
    print("multistage bf16 %s: losses %.3e  stage1 max %.3e  stage2 (teacher-forced) max %.3e rms %.3e  [end-to-end stage2 max %.3e; "
          "mask differs at %d pixels, none of them radar returns]" % (geom, e_loss, e1, e2, r2, e2e, flips))
    assert e_loss < 5e-4 and e1 < 0.1 and e2 < 0.1 and r2 < 4e-2      # measured: 4e-4 / 2.9e-2 / 2.9e-2 / 2.1e-2 and 1.5e-4 / 3.5e-2 / 3.4e-2 / 2.3e-2
    if small:
        names = [n for n, _ in om.named_parameters()]
        go = np.array([p.grad.double().norm().item() for p in om.parameters()])
        gg = np.array([(i0 - p.detach()).double().norm().item() for i0, p in zip(init, hm.parameters())])
        e_norm = np.abs(go - gg).max() / go.max()
        print("  gradient norms: worst %.3e (%s)" % (e_norm, names[int(np.abs(go - gg).argmax())]))
        # (stage1.conv1.weight: reached through stage 2's depth stem and the whole stage-2 backward.  The emulated oracle's own
        #  norms move by up to 1.1e-1 there under a 1e-6 weight perturbation, 3.4e-2 elsewhere --
        #  tests/test_conditioning.py::test_bf16_storage_gradient_norm_floor; measured here 3.5e-2 ... 9e-2 over kernel revisions)
        body = [i for i, n in enumerate(names) if not n.endswith(("conv1.weight", "conv1_depth.weight")) or ".layer" in n]
        assert e_norm < 0.2 and np.abs(go[body] - gg[body]).max() / go.max() < 7e-2
        assert abs(gg[0] - go[0]) < 1e-3 * max(go[0], 1e-6) + 1e-6 and abs(gg[1] - go[1]) < 1e-3 * max(go[1], 1e-6) + 1e-6


# ------------------------------------------------------------------------------------------------ plain multistage
@pytest.mark.parametrize("operands", [pytest.param("fp32", marks=pytest.mark.slow), "split"])
def test_plain_multistage_step_vs_oracle(operands):
    """--arch resnet18_multistage (main.py:431-438): loss = d1 + d2, no uncertainty weights, no smoothness term.  Both plans of the fused
    step; the fp32-MFMA plan keeps the tighter third-step bar it had before the split plan became the default."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    args, hm, hw_, om, ow = _multistage_pair(h, w, arch="resnet18_multistage")
    assert hw_ is None and ow is None and "w_stage1" not in dict(hm.named_parameters())
    opt = torch.optim.SGD(om.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    crit = otrain.make_criterion(args.arch)
    ts = HipTrainStep(hm, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=None, operands=operands)
    for it in range(3):
        x, t = make_batch(b, h, w, 700 + it, ref_pixels=h * w)
        lo, po, ex = otrain.train_step(args.arch, om, crit, opt, x, t, None)
        lg, pred = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        # step 0 is a forward parity check (measured 0 .. 1e-7); from then on the two trajectories amplify their rounding differences: five
        # arithmetic variants of this library (split / fp32-MFMA plans, stem / 16-channel / head kernel forms) sit 0.9e-4 .. 5e-4 from the oracle
        # at step 1 and 0.95e-3 .. 2.3e-3 at step 2 -- each other's distance as much as the oracle's
        bar = ((1e-5, 2e-3, 2.5e-3) if operands == "fp32" else (1e-5, 2e-3, 5e-3))[it]
        print("plain multistage [%s] step %d: loss rel err %.3e (bar %.1e)" % (operands, it, abs(lg.item() - lo.item()) / abs(lo.item()), bar))
        assert abs(lg.item() - lo.item()) / abs(lo.item()) < bar, (it, lg.item(), lo.item())
        po_ = np.array([p.double().norm().item() for p in om.parameters()])
        pg_ = np.array([p.double().norm().item() for p in hm.parameters()])
        if it == 0:
            assert rel(_t(pred), _t(po)) < 1e-3
            assert abs(lg.item() - (ex["d1"].item() + ex["d2"].item())) < 1e-4 * abs(lg.item())
            assert np.abs(po_ - pg_).max() / po_.max() < 1e-3       # one SGD step from identical states
    # three steps: the gradients are e (2.7x) larger than with the uncertainty weights e^-1, and so is the chaotic divergence of the
    # trajectories (tests/test_conditioning.py) -- 5e-3 in the uncertainty test corresponds to 1.5e-2 here
    assert np.abs(po_ - pg_).max() / po_.max() < 1.5e-2


# ------------------------------------------------------------------------------------------------ -c l2
def test_masked_mse_loss_vs_golden_and_oracle():
    """MaskedMSELoss (evaluation/criteria_new.py:31-41): the reference-generated value in units.npz, the gradient against torch
    autograd of the oracle's restatement, the NaN of an all-invalid target, and the fused step with criterion='l2'."""
    from oracle.criteria import MaskedMSELoss as OL2
    from radar_depth_amd.evaluation.criteria_new import MaskedMSELoss
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    want = np.load(os.path.join(GOLD, "units.npz"))
    pred = torch.tensor(want["l1/pred"]).cuda().requires_grad_(True)
    target = torch.tensor(want["l1/target"]).cuda()
    loss = MaskedMSELoss()(pred, target)
    assert abs(loss.item() - want["l2/loss"][0]) / want["l2/loss"][0] < 1e-6
    (3.0 * loss).backward()
    pc = torch.tensor(want["l1/pred"]).requires_grad_(True)
    (3.0 * OL2()(pc, torch.tensor(want["l1/target"]))).backward()
    assert rel(_t(pred.grad), _t(pc.grad)) < 1e-6
    assert torch.isnan(MaskedMSELoss()(pred.detach(), torch.zeros_like(target)))
    b, h, w = 2, 97, 161
    m, o = _latefusion_pair(h, w)
    opt = torch.optim.SGD(o.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    ts = HipTrainStep(m, b, h, w, criterion="l2")
    for it in range(2):
        x, t = make_batch(b, h, w, 40 + it, ref_pixels=h * w)
        lo = OL2()(o(x), t)
        opt.zero_grad()
        lo.backward()
        opt.step()
        lg, _ = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        assert abs(lg.item() - lo.item()) / lo.item() < 2e-3, (it, lg.item(), lo.item())


# ------------------------------------------------------------------------------------------------ layer-level pins
class _Holder:
    pass


def _call(mod, x_np, gy_np, operands="split"):
    """module(x); y.backward(gy) -- the sub-module called on its own, as the reference's can be; operands: the arithmetic of its
    eager plan (module.operands: "split" -- the default -- or "fp32")."""
    mod = mod.cuda().train()
    mod.operands = operands
    x = torch.tensor(x_np).cuda().requires_grad_(True)
    y = mod(x)
    y.backward(torch.tensor(gy_np).cuda())
    torch.cuda.synchronize()
    return y, x.grad


@pytest.mark.parametrize("operands", ["split", "fp32"])
def test_upproj_module_fwd_bwd_vs_golden(operands):
    """One UpProjModule(32) (models.py:181-209: unpool -> 5x5 ‖ 5x5 -> BN/ReLU -> 3x3 -> BN -> add -> ReLU) called STAND-ALONE
    (module(x); y.backward(gy): a one-module plan built from the network plan's own builders -- 4-phase zero-skipping conv, fused BN
    statistics, joined BN backward, 25-tap dgrad, per-phase wgrad) against the reference's module output, input gradient and all 9
    parameter gradients.  Tolerance 1e-4 of each tensor's max (the kernels are exact-fp32 fmaf chains; differences are summation order)."""
    from radar_depth_amd.model.models import UpProj
    from radar_depth_amd.synthetic import procedural_fill_
    want = np.load(os.path.join(GOLD, "upproj_module.npz"))
    mod = UpProj.UpProjModule(32)
    procedural_fill_(mod)
    y, dx = _call(mod, want["x"], want["gy"], operands)
    assert [pl.split for pl in mod.__dict__["_module_plans"].values()] == [operands == "split"]
    assert rel(_t(y), want["y"]) < 1e-4
    assert rel(_t(dx), want["gx"]) < 1e-4
    checked = 0
    for name, p in mod.named_parameters():
        assert rel(_t(p.grad), want["grad/" + name]) < 1e-4, name
        checked += 1
    assert checked == 9


@pytest.mark.parametrize("operands", ["split", "fp32"])
@pytest.mark.parametrize("tag,cin,cout,stride", [("id", 32, 32, 1), ("ds", 32, 64, 2), ("id16", 16, 16, 1)])
def test_basic_block_fwd_bwd_vs_golden(tag, cin, cout, stride, operands):
    """The reference's own BasicBlock (models.py:75-112), identity-residual and stride-2 + 1x1-downsample forms, called stand-alone:
    output, input gradient (residual + conv paths summed in the dgrad epilogue) and every parameter gradient vs vectors generated from
    the reference; 1e-4 of each tensor's max.  id16 is the depth encoder's layer1 block (BasicBlock(16, 16), models.py:567) on a
    19x37 map: the 16-channel kernels with their BatchNorm joins behind a reference-generated pin."""
    from radar_depth_amd.model.models import BasicBlock, _conv
    from radar_depth_amd.synthetic import procedural_fill_
    want = np.load(os.path.join(GOLD, "basic_block16.npz" if tag == "id16" else "basic_block.npz"))
    down = None
    if stride != 1 or cin != cout:
        down = torch.nn.Sequential(_conv(cin, cout, 1, stride, pad=0), torch.nn.BatchNorm2d(cout))
    mod = BasicBlock(cin, cout, stride, down)
    procedural_fill_(mod)
    y, dx = _call(mod, want[tag + "/x"], want[tag + "/gy"], operands)
    assert rel(_t(y), want[tag + "/y"]) < 1e-4
    assert rel(_t(dx), want[tag + "/gx"]) < 1e-4
    for name, p in mod.named_parameters():
        assert rel(_t(p.grad), want[tag + "/grad/" + name]) < 1e-4, name


def test_submodules_standalone_vs_oracle():
    """The other stand-alone forms of the contract's sub-modules against the CPU oracle's modules with the same parameters:
    the whole UpProj decoder (models.py:210-216, four modules in a row) forward + backward in training mode, a BasicBlock and an
    UpProjModule in EVAL mode (running statistics, BatchNorm folded into the convolutions), Unpool (models.py:13-27) with its
    gradient; and a training-mode forward of a sub-module must update its BatchNorm running statistics like the oracle's."""
    from oracle import models as om
    from radar_depth_amd.model import models as hm
    from radar_depth_amd.synthetic import procedural_fill_
    torch.manual_seed(3)
    # decoder, training mode
    dec, odec = hm.UpProj(256), om.UpProj(256)          # (the networks' decoder: modules of 256 / 128 / 64 / 32 input channels)
    procedural_fill_(dec)
    procedural_fill_(odec)
    x = torch.randn(2, 256, 5, 7)
    gy = torch.randn(2, 16, 80, 112)
    xo = x.clone().requires_grad_(True)
    yo = odec.train()(xo)
    yo.backward(gy)
    y, dx = _call(dec, x.numpy(), gy.numpy())
    assert rel(_t(y), _t(yo)) < 1e-4 and rel(_t(dx), _t(xo.grad)) < 2e-4
    for (n, p), (_, q) in zip(dec.named_parameters(), odec.named_parameters()):
        assert rel(_t(p.grad), _t(q.grad)) < 2e-4, n
    for k in ("layer1.upper_branch.batchnorm1.running_mean", "layer4.bottom_branch.batchnorm.running_var"):
        assert rel(_t(dec.state_dict()[k]), _t(odec.state_dict()[k])) < 1e-4, k
    # eval mode: BasicBlock (stride 2 + downsample) and UpProjModule, after the statistics moved
    down = torch.nn.Sequential(hm._conv(32, 64, 1, 2, pad=0), torch.nn.BatchNorm2d(64))
    odown = torch.nn.Sequential(om._conv(32, 64, 1, 2, pad=0), torch.nn.BatchNorm2d(64))
    for mod, omod, shape in ((hm.BasicBlock(32, 64, 2, down), om.BasicBlock(32, 64, 2, odown), (2, 32, 11, 13)),
                             (hm.UpProj.UpProjModule(32), om.UpProj.UpProjModule(32), (2, 32, 7, 9))):
        procedural_fill_(mod)
        procedural_fill_(omod)
        for bn, obn in zip([m for m in mod.modules() if isinstance(m, torch.nn.BatchNorm2d)], [m for m in omod.modules() if isinstance(m, torch.nn.BatchNorm2d)]):
            rm, rv = torch.randn(bn.num_features) * 0.1, torch.rand(bn.num_features) + 0.5
            for b_ in (bn, obn):
                b_.running_mean.copy_(rm)
                b_.running_var.copy_(rv)
        xe = torch.randn(*shape)
        with torch.no_grad():
            ye = mod.cuda().eval()(xe.cuda())
            yoe = omod.eval()(xe)
        assert rel(_t(ye), _t(yoe)) < 1e-4, type(mod).__name__
    # Unpool
    up, oup = hm.Unpool(8), om.Unpool(8)
    xu = torch.randn(2, 8, 5, 6)
    xg = xu.clone().cuda().requires_grad_(True)
    yu = up(xg)
    gu = torch.randn_like(yu)
    yu.backward(gu)
    xog = xu.clone().requires_grad_(True)
    you = oup(xog)
    you.backward(gu.cpu())
    assert torch.equal(yu.detach().cpu(), you.detach()) and torch.equal(xg.grad.cpu(), xog.grad)


def test_submodule_call_between_steps_keeps_the_network_arena():
    """model.layer1[0](x) / model.decoder(x) between two fused steps must run on the NETWORK's arenas -- the sub-module is not
    re-homed into an arena of its own (the fused step would keep training the old slots while state_dict() reads frozen copies)."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m, _ = _latefusion_pair(h, w)
    ts = HipTrainStep(m, b, h, w)
    st = m._ensure_arenas()
    x, t = [v.cuda() for v in make_batch(b, h, w, 3, ref_pixels=h * w)]
    ts.step(x, t)
    blk, dec = m.layer1[0], m.decoder
    assert blk._arena_root() is m and dec.layer2._arena_root() is m
    m.zero_grad()
    xb = torch.randn(2, 64, 25, 41, device="cuda", requires_grad=True)
    blk(xb).square().mean().backward()                    # training-mode stand-alone call + backward
    with torch.no_grad():
        dec.eval()(torch.randn(1, 256, 4, 6, device="cuda"))
        dec.train()
    torch.cuda.synchronize()
    own = {id(p) for p in blk.parameters()}
    assert all((p.grad is not None) == (id(p) in own) for p in m.parameters())      # only the block's parameters received gradients
    assert "_arena_state" not in blk.__dict__ and "_arena_state" not in dec.__dict__
    assert m._ensure_arenas() is st and [p.data_ptr() for p in st["params"]] == st["ptrs"]
    before = st["arena"].clone()
    ts.step(x, t)                                         # no "arena was rebuilt" error, and the step trains what state_dict() reads
    torch.cuda.synchronize()
    assert not torch.equal(before, st["arena"])
    off = 0
    sd = m.state_dict()
    for name, p in m.named_parameters():
        assert torch.equal(sd[name].reshape(-1), st["arena"][off:off + p.numel()]), name
        off += (p.numel() + 3) // 4 * 4
    # a module built on its own and re-homed behind a live step IS noticed (every pointer is re-checked after any arena build)
    from radar_depth_amd.model.models import BasicBlock
    other = BasicBlock(16, 16).cuda()
    other(torch.randn(1, 16, 9, 9, device="cuda"))          # builds an arena of its own: bumps the global re-home counter, moves nothing of m
    ts.step(x, t)
    m.layer2[1].conv1.weight.data = m.layer2[1].conv1.weight.data.clone()      # a middle parameter leaves the arena
    other.__dict__.pop("_arena_state")
    other(torch.randn(1, 16, 9, 9, device="cuda"))
    with pytest.raises(RuntimeError):
        ts.step(x, t)


# ------------------------------------------------------------------------------------------------ robustness
def test_step_rejects_ragged_batch_and_stale_autograd():
    from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m, _ = _latefusion_pair(h, w)
    ts = HipTrainStep(m, b, h, w)
    x, t = make_batch(1, h, w, 3, ref_pixels=h * w)
    with pytest.raises(ValueError):
        ts.step(x.cuda(), t.cuda())                       # a last batch of 1 must not be broadcast over the static buffers
    x, t = make_batch(b, h, w, 3, ref_pixels=h * w)
    y1 = m(x.cuda())
    y2 = m(x.cuda())                                      # second training-mode forward overwrites the saved activations
    with pytest.raises(RuntimeError):
        MaskedL1Loss()(y1, t.cuda()).backward()
    MaskedL1Loss()(y2, t.cuda()).backward()               # the latest forward is still differentiable


def test_optimizer_state_round_trip():
    """HipTrainStep.state_dict() is torch.optim.SGD's layout (per-parameter momentum_buffer): loading it into torch's SGD and
    into a fresh HipTrainStep continues the same trajectory."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m1, _ = _latefusion_pair(h, w)
    m2, _ = _latefusion_pair(h, w)
    t1 = HipTrainStep(m1, b, h, w)
    batches = [tuple(v.cuda() for v in make_batch(b, h, w, 10 + i, ref_pixels=h * w)) for i in range(3)]
    for x, t in batches[:2]:
        t1.step(x, t)
    sd = t1.state_dict()
    assert set(sd) == {"state", "param_groups"} and len(sd["state"]) == len(list(m1.parameters()))
    opt = torch.optim.SGD(m1.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    opt.load_state_dict(sd)                               # torch accepts the layout
    m2.load_state_dict(m1.state_dict())
    t2 = HipTrainStep(m2, b, h, w)
    t2.load_state_dict(sd)
    l1, _ = t1.step(*batches[2])
    l2, _ = t2.step(*batches[2])
    torch.cuda.synchronize()
    assert l1.item() == l2.item()
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(p, q)


# ------------------------------------------------------------------------------------------------ native RCCL communicator
def test_native_rccl_comm_single_rank(tmp_path, monkeypatch):
    """rd_comm_unique_id / rd_comm_init / rd_allreduce_bucket / rd_broadcast (the C ABI's own RCCL communicator)
    on a 1-rank communicator: all-reduce is the identity, and the data-parallel step (bucketed all-reduces on the communication
    stream, event-chained per backward segment, 1/world in SGD) reproduces the single-GPU step bit for bit -- latefusion
    (4 buckets) and multistage (8 buckets)."""
    import ctypes as C
    from radar_depth_amd import comm
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m_ref, _ = _latefusion_pair(h, w)
    ts_ref = HipTrainStep(m_ref, b, h, w)
    args, hm_ref, hw_ref, _, _ = _multistage_pair(h, w)
    ms_ref = HipTrainStep(hm_ref, b, h, w, loss_weights=hw_ref)
    comm.init_from_file(str(tmp_path / "rccl_token"), 0, 1)
    try:
        assert comm.world() == 1 and comm.rank() == 0
        v = torch.arange(1000, dtype=torch.float32, device="cuda")
        cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        comm.allreduce_(v, cur, 100, 900)
        comm.broadcast_(v, cur, 0)
        torch.cuda.synchronize()
        assert torch.equal(v.cpu(), torch.arange(1000, dtype=torch.float32))
        monkeypatch.setenv("RD_FORCE_DP", "1")
        m, _ = _latefusion_pair(h, w)
        ts = HipTrainStep(m, b, h, w)
        _, hm, hw_, _, _ = _multistage_pair(h, w)
        ms = HipTrainStep(hm, b, h, w, loss_weights=hw_)
        assert ts.comm == "rccl" and ts.dp and len(ts._buckets) == 4 and ms.dp and len(ms._buckets) == 8
        # the buckets partition the gradient arena exactly
        for step_ in (ts, ms):
            covered = sorted(sl for bk in step_._buckets for sl in bk)
            assert covered[0][0] == 0 and covered[-1][1] == step_.st["total"]
            assert all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))
        for it in range(3):
            x, t = make_batch(b, h, w, 300 + it, ref_pixels=h * w)
            l0, _ = ts_ref.step(x.cuda(), t.cuda())
            l1, _ = ts.step(x.cuda(), t.cuda())
            k0, _ = ms_ref.step(x.cuda(), t.cuda())
            k1, _ = ms.step(x.cuda(), t.cuda())
            torch.cuda.synchronize()
            assert l0.item() == l1.item() and k0.item() == k1.item()
        for p, q in zip(m_ref.parameters(), m.parameters()):
            assert torch.equal(p, q)
        for p, q in zip(hm_ref.parameters(), hm.parameters()):
            assert torch.equal(p, q)
    finally:
        comm.destroy()


def _two_rank_worker(rank, world, token_path, out_path):
    import torch
    from radar_depth_amd import comm
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    torch.cuda.set_device(rank)
    comm.init_from_file(token_path, rank, world, job_id="two-rank-test")
    b, h, w = 2, 97, 161
    torch.manual_seed(100 + rank)                      # deliberately different seeds: rank 0's state must win
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    if rank == 0:
        procedural_fill_(m)
    m = m.cuda()
    ts = HipTrainStep(m, b, h, w)
    x, t = make_batch(b, h, w, 1234 + 1000 * rank, ref_pixels=h * w)
    ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    torch.save([p.detach().cpu() for p in m.parameters()], "%s.%d" % (out_path, rank))
    comm.destroy()


def test_native_rccl_two_ranks(tmp_path):
    """Two processes, two GPUs, RCCL over xGMI through the C ABI: replicas start from rank 0's state (broadcast) although they
    were seeded differently, see different batches, and end with identical parameters equal to a single process that averages
    the two batches' gradients itself.  Skipped on a 1-GPU box."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch.multiprocessing as mp
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    token, out = str(tmp_path / "tok"), str(tmp_path / "params")
    mp.start_processes(_two_rank_worker, args=(2, token, out), nprocs=2, start_method="spawn")
    p0, p1 = torch.load(out + ".0"), torch.load(out + ".1")
    for a, c in zip(p0, p1):
        assert torch.equal(a, c)
    # reference: one process, gradient = mean of the two ranks' gradients (per-rank BatchNorm statistics)
    b, h, w = 2, 97, 161
    ma, _ = _latefusion_pair(h, w)
    mb, _ = _latefusion_pair(h, w)
    ta = HipTrainStep(ma, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0)      # gradient probes (no update)
    tb = HipTrainStep(mb, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0)
    xa, ta_ = make_batch(b, h, w, 1234, ref_pixels=h * w)
    xb, tb_ = make_batch(b, h, w, 2234, ref_pixels=h * w)
    ta.step(xa.cuda(), ta_.cuda())
    tb.step(xb.cuda(), tb_.cuda())
    g = 0.5 * (ta.st["grads"] + tb.st["grads"])
    # one data-parallel step from rank 0's state: p1 = p0 - lr * (g + wd * p0)  (the momentum buffer starts at the gradient)
    want = ta.st["arena"] - 0.01 * (g + 1e-4 * ta.st["arena"])
    off = 0
    for a in p0:
        ref = want[off:off + a.numel()].view(a.shape).cpu()
        assert (a - ref).abs().max().item() <= 1e-6 * max(ref.abs().max().item(), 1e-3)
        off += (a.numel() + 3) // 4 * 4


def test_training_loop_checkpoint_and_resume(tmp_path):
    """The reference's loop shape end to end (tools/train_synthetic.py: parse_command -> create_model -> fused steps -> LR schedule
    -> on-device metrics -> validate() via the inference graph -> .pth.tar with optimizer state -> --resume): a run of two epochs
    equals one epoch + resume + one epoch, bit for bit (parameters and momentum restored, LR schedule continued)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_synthetic", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                              "tools", "train_synthetic.py"))
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    common = ["-a", "resnet18_latefusion", "-d", "upproj", "-m", "rgbd", "--data", "nuscenes", "-b", "2", "--no-pretrain",
              "--steps-per-epoch", "3", "--height", "97", "--width", "161"]
    torch.manual_seed(5)
    m_full, _ = ts.main(common + ["--epochs", "2", "--output", str(tmp_path / "full")])
    torch.manual_seed(5)
    ts.main(common + ["--epochs", "1", "--output", str(tmp_path / "part")])
    torch.manual_seed(77)                                  # a different init: everything must come from the checkpoint
    m_res, _ = ts.main(common + ["--epochs", "2", "--output", str(tmp_path / "part"), "--resume", str(tmp_path / "part" / "checkpoint-0.pth.tar")])
    for (k, a), (_, c) in zip(m_full.state_dict().items(), m_res.state_dict().items()):
        assert torch.equal(a, c), k
    from radar_depth_amd import utils
    ck = utils.load_checkpoint(str(tmp_path / "part" / "checkpoint-1.pth.tar"))
    assert ck["epoch"] == 1 and ck["arch"] == "resnet18_latefusion" and len(ck["optimizer_state_dict"]["state"]) == 163
    assert ck["args"].decoder == "upproj" and ck["best_result"].rmse < float("inf")


@pytest.mark.slow
def test_autotuned_plans_keep_parity_and_do_not_disturb_existing_plans():
    """The gconv plan tuner (radar_depth_amd/autotune.py): (1) a plan built before tuning keeps working -- descriptors that were
    already planned are never re-pinned under it; (2) a model built with autotune=True at a geometry nobody planned yet reproduces
    the oracle's forward / loss / gradient norms like the heuristic plans do; (3) every candidate the tuner timed agreed with the
    reference result (it warns and rejects otherwise)."""
    import warnings
    from oracle.criteria import MaskedL1Loss as OL1
    from radar_depth_amd import autotune
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    m0, _ = _latefusion_pair(h, w)
    t0 = HipTrainStep(m0, b, h, w)                        # heuristic plans at this geometry: now in use
    x, t = make_batch(b, h, w, 11, ref_pixels=h * w)
    l_before, _ = t0.step(x.cuda(), t.cuda())
    m1, _ = _latefusion_pair(h, w)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                    # a rejected candidate would warn
        t1 = HipTrainStep(m1, b, h, w, autotune=True)     # same descriptors: nothing may be re-pinned
        l_same, _ = t1.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        assert l_before.item() == l_same.item()           # same plans -> bit-identical
        # a fresh geometry: tuned from scratch
        b2, h2, w2 = 3, 113, 145
        n0 = autotune.summary()[0]
        m2, o2 = _latefusion_pair(h2, w2)
        x2, tg2 = make_batch(b2, h2, w2, 12, ref_pixels=h2 * w2)
        yo = o2(x2)
        lo = OL1()(yo, tg2)
        lo.backward()
        init = [p.detach().clone() for p in m2.parameters()]
        t2 = HipTrainStep(m2, b2, h2, w2, lr=1.0, momentum=0.0, weight_decay=0.0, autotune=True)
        loss, pred = t2.step(x2.cuda(), tg2.cuda())
        torch.cuda.synchronize()
    n1, heur_us, tuned_us = autotune.summary()
    assert n1 > n0 and tuned_us <= heur_us
    assert rel(_t(pred), _t(yo)) < 1e-3 and abs(loss.item() - lo.item()) / lo.item() < 1e-4
    go = np.array([p.grad.double().norm().item() for p in o2.parameters()])
    gg = np.array([(i0 - p.detach()).double().norm().item() for i0, p in zip(init, m2.parameters())])
    assert np.abs(go - gg).max() / go.max() < 2e-2
    l_after, _ = t0.step(x.cuda(), t.cuda())              # the first model's plan still runs (and still agrees with its twin)
    l_twin, _ = t1.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    assert l_after.item() == l_twin.item() and np.isfinite(l_after.item())


@pytest.mark.parametrize("order", ["table_then_tuner", "tuner_then_table"])
def test_committed_plans_survive_a_later_tuner_at_a_table_geometry(order):
    """at a geometry the offline-tuned table has entries for (b=16, 450x800), a plan whose buffers were sized on a pinned plan
    must not be re-pinned by a later HipTrainStep(autotune=True) in the same process -- nor a tuned plan by a later table lookup.  Both
    orders: the second model's step must leave the first model's step bit-reproducible, and every gconv plan the first one committed
    must still be the plan the library reports."""
    import ctypes as C
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 16, 450, 800
    x, t = make_batch(b, h, w, 77)
    x, t = x.cuda(), t.cuda()
    first_tuned = order == "tuner_then_table"
    m1, _ = _latefusion_pair(h, w)
    t1 = HipTrainStep(m1, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0, operands="fp32", autotune=first_tuned)
    L = t1.L

    def plans(ts):
        out = {}
        for name, (kind, d) in ts.plan.meta.items():
            if kind == "gconv":
                info = (C.c_int32 * 10)()
                assert L.rd_gconv_plan_info(C.byref(d), info) == 0
                out[name] = tuple(info)
                assert L.rd_gconv_plan_state(C.byref(d), 1) == 2, name          # committed: in use
        return out
    before = plans(t1)
    l1, _ = t1.step(x, t)
    torch.cuda.synchronize()
    ref = l1.item()
    m2, _ = _latefusion_pair(h, w)
    t2 = HipTrainStep(m2, b, h, w, lr=0.0, momentum=0.0, weight_decay=0.0, operands="fp32", autotune=not first_tuned)
    l2, _ = t2.step(x, t)
    torch.cuda.synchronize()
    assert plans(t1) == before, "a later plan build re-pinned a committed plan"
    assert plans(t2) == before, "the same descriptors must resolve to the same committed plans"
    l1b, _ = t1.step(x, t)                       # lr = 0: the step is a pure function of (parameters, batch)
    torch.cuda.synchronize()
    assert l1b.item() == ref and l2.item() == ref
