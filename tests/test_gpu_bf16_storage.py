"""bf16 STORAGE path (BASELINE.json configs 3 / 5): NHWC activations and their gradients live in HBM as bf16, convolutions run on
v_mfma_f32_32x32x16_bf16 with fp32 accumulation, BatchNorm statistics / losses / parameters / parameter gradients / SGD stay fp32.

Kernel level: every storage-typed entry point (`*_t` with RD_DTYPE_BF16) against a float64 torch evaluation of the SAME bf16 inputs;
outputs are compared with the reference rounded to bf16 -- one bf16 ulp (2^-8 relative) is the bar, because a 1e-7 fp32
summation-order difference can move a value across a rounding boundary.
Step level: the CPU oracle with the same rounding points (tensors rounded where the HIP plan stores them, conv operands and
gradients bf16, fp32 everywhere else), stated tolerances in each test."""
import ctypes as C
import importlib.util
import os
import types

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BF16 = 1          # RD_DTYPE_BF16
ULP = 2.0 ** -8   # one bf16 ulp, relative (8 significand bits incl. the implicit one -> spacing 2^-7 .. 2^-8 of the value)


def _bfmod():
    spec = importlib.util.spec_from_file_location("_bf16_tests", os.path.join(HERE, "test_gpu_bf16.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _ulp_err(got_bf16, ref64):
    """max |got - ref| in units of the local bf16 spacing of the reference (plus a floor of 1e-3 of the tensor's max)."""
    got = got_bf16.double().cpu()
    scale = torch.maximum(ref64.abs(), torch.full_like(ref64, 1e-3 * ref64.abs().max().item()))
    return ((got - ref64).abs() / (scale * ULP)).max().item()


def _nhwc16(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


@pytest.mark.parametrize("cfg", [(2, 32, 64, 3, 1, 23, 31), (1, 64, 32, 3, 2, 30, 41), (2, 128, 48, 1, 1, 9, 14), (1, 16, 16, 3, 1, 40, 70),
                                 (2, 64, 128, 1, 2, 17, 19), (1, 256, 64, 3, 1, 15, 25)])
def test_gconv_bf16_storage_forward_dgrad(cfg):
    """rd_gconv_bf16_t(RD_DTYPE_BF16): bf16 tensors in, bf16 tensors out, fp32 BatchNorm partial sums from the accumulators."""
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    n, ci, co, k, s, h, w = cfg
    L = lib()
    g = torch.Generator().manual_seed(sum(cfg))
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    ref = F.conv2d(x.double(), _bf(wt).double(), None, s, k // 2)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, k // 2)
    xs = _nhwc16(x)
    wp = ops.pack_weights_bf16(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), dtype=torch.bfloat16, device="cuda")
    stat = torch.full((L.rd_gconv_bf16_stat_tiles_t(BF16, C.byref(d)), 2, co), float("nan"), device="cuda")
    check(L.rd_gconv_bf16_t(BF16, C.byref(d), ptr(xs), ptr(wp), ptr(out), None, 0, 0, None, 0, ptr(stat), current_stream()), "gconv_bf16_t")
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2)
    assert not torch.isnan(got.float()).any()
    assert _ulp_err(got, ref) <= 1.01, _ulp_err(got, ref)
    s_ = stat.double().sum(0).cpu()                      # statistics of the UNROUNDED fp32 results
    assert ((s_[0] - ref.sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).sqrt().max()).item() < 1e-4
    assert ((s_[1] - (ref ** 2).sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).max()).item() < 1e-4
    if co % 16 == 0:
        gy = _bf(torch.randn(ref.shape, generator=g))
        add = _bf(torch.randn(n, ci, h, w, generator=g))
        dref = torch.nn.grad.conv2d_input((n, ci, h, w), _bf(wt).double(), gy.double(), s, k // 2) + add.double()
        dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, k // 2)
        dx = torch.zeros(n, h, w, ci, dtype=torch.bfloat16, device="cuda") if zero_fill else \
            torch.full((n, h, w, ci), float("nan"), dtype=torch.bfloat16, device="cuda")
        wd = ops.pack_weights_bf16(wt.cuda(), transpose=True)
        adds = None if zero_fill else _nhwc16(add)
        if zero_fill:
            dref = dref - add.double()
        gys = _nhwc16(gy)
        check(L.rd_gconv_bf16_t(BF16, C.byref(dd), ptr(gys), ptr(wd), ptr(dx), None, 0, 0, ptr(adds), ci if adds is not None else 0, None,
                                current_stream()), "gconv_bf16_t dgrad")
        torch.cuda.synchronize()
        assert _ulp_err(dx.permute(0, 3, 1, 2), dref) <= 1.01


@pytest.mark.parametrize("grid_cap", [0, 1, 5])
@pytest.mark.parametrize("cfg", [(2, 64, 64, 23, 31, "conv"), (3, 32, 32, 57, 100, "conv"), (1, 128, 64, 29, 50, "conv"), (2, 64, 96, 13, 29, "conv"),
                                 (1, 256, 256, 15, 25, "conv"), (2, 64, 64, 9, 13, "upproj"), (1, 32, 32, 30, 50, "upproj"), (2, 128, 128, 15, 25, "upproj"),
                                 (2, 64, 64, 3, 5, "conv")])
def test_gconv_bf16p_persistent_kernel(cfg, grid_cap, monkeypatch):
    """csrc/gconv_bf16p.hip (the persistent pipelined kernel rd_gconv_bf16_t dispatches bf16 tensors with a unit-stride input and 4..9
    taps per phase to): 3x3 forward with BatchNorm partial sums, its input gradient with a residual addend, and the four-phase UpProj
    forward, against fp64 convolutions of the same bf16 operands: every output within one bf16 ulp, statistics within 1e-4.
    grid_cap 1 / 5: the whole tile list walked by one / five workgroups (the cross-tile pipeline: ring slots, prefetch across the
    tile boundary, scratch reuse); 0: the launch's own grid.  Ragged tiles, edge tiles and images smaller than a tile included."""
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    n, ci, co, h, w, kind = cfg
    L = lib()
    prev = L.rd_gconv_bf16p_plan_all(1)
    try:
        _bf16p_case(cfg, grid_cap, monkeypatch)
    finally:
        L.rd_gconv_bf16p_plan_all(prev)


def _bf16p_case(cfg, grid_cap, monkeypatch):
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    n, ci, co, h, w, kind = cfg
    L = lib()
    if grid_cap:
        monkeypatch.setenv("RD_GCONV_BF16P_GRID", str(grid_cap))
    else:
        monkeypatch.delenv("RD_GCONV_BF16P_GRID", raising=False)
    g = torch.Generator().manual_seed(sum(cfg[:5]) + grid_cap)
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    if kind == "upproj":
        wt = torch.randn(co, ci, 5, 5, generator=g) * (2.0 / (9 * ci)) ** 0.5
        up = torch.zeros(n, ci, 2 * h, 2 * w)
        up[:, :, ::2, ::2] = x
        ref = F.conv2d(up.double(), _bf(wt).double(), None, 1, 2)
        d = cd.upproj_fwd(n, h, w, ci, co)
    else:
        wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
        ref = F.conv2d(x.double(), _bf(wt).double(), None, 1, 1)
        d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    info = (C.c_int32 * 8)()
    check(L.rd_gconv_bf16_plan_info_t(BF16, C.byref(d), info), "plan_info_t")
    assert info[2] >= 2000, "this shape must be served by the persistent kernel"
    xs = _nhwc16(x)
    wp = ops.pack_weights_bf16(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), dtype=torch.bfloat16, device="cuda")
    stat = torch.full((L.rd_gconv_bf16_stat_tiles_t(BF16, C.byref(d)), 2, co), float("nan"), device="cuda")
    for rep in range(2):        # (twice: the second launch must not depend on what the first left in the LDS / the slot table cache;
        if rep == 1:            #  and with every CU's LDS NaN-filled first: masked slots, unstaged patch rows and ring slots are never consumed)
            check(L.rd_debug_poison_lds(current_stream()), "poison_lds")
        check(L.rd_gconv_bf16_t(BF16, C.byref(d), ptr(xs), ptr(wp), ptr(out), None, 0, 0, None, 0, ptr(stat), current_stream()), "gconv_bf16_t")
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2)
    assert not torch.isnan(got.float()).any()
    assert _ulp_err(got, ref) <= 1.01, _ulp_err(got, ref)
    s_ = stat.double().sum(0).cpu()
    assert ((s_[0] - ref.sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).sqrt().max()).item() < 1e-4
    assert ((s_[1] - (ref ** 2).sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).max()).item() < 1e-4
    if kind == "conv":
        gy = _bf(torch.randn(ref.shape, generator=g))
        add = _bf(torch.randn(n, ci, h, w, generator=g))
        dref = torch.nn.grad.conv2d_input((n, ci, h, w), _bf(wt).double(), gy.double(), 1, 1) + add.double()
        dd, _ = cd.conv_dgrad(n, h, w, ci, co, 3, 1, 1)
        check(L.rd_gconv_bf16_plan_info_t(BF16, C.byref(dd), info), "plan_info_t")
        assert info[2] >= 2000
        dx = torch.full((n, h, w, ci), float("nan"), dtype=torch.bfloat16, device="cuda")
        wd = ops.pack_weights_bf16(wt.cuda(), transpose=True)
        adds, gys = _nhwc16(add), _nhwc16(gy)
        check(L.rd_debug_poison_lds(current_stream()), "poison_lds")
        check(L.rd_gconv_bf16_t(BF16, C.byref(dd), ptr(gys), ptr(wd), ptr(dx), None, 0, 0, ptr(adds), ci, None, current_stream()), "gconv_bf16_t dgrad")
        torch.cuda.synchronize()
        assert _ulp_err(dx.permute(0, 3, 1, 2), dref) <= 1.01


@pytest.mark.parametrize("cfg", [(2, 64, 64, 3, 1, 23, 31), (1, 32, 64, 3, 2, 30, 41), (2, 128, 64, 1, 1, 9, 14), (2, 16, 16, 3, 1, 40, 70),
                                 (1, 16, 32, 1, 2, 33, 35), (2, 64, 32, 5, 1, 12, 10)])
def test_wgrad_bf16_storage(cfg):
    """rd_wgrad_bf16_t(RD_DTYPE_BF16): bf16 x / dy (no conversion in the staging waves) -> fp32 weight gradient; incl. the
    16-channel layers, which bf16-storage plans also route here, and the UpProj 5x5 (four phases)."""
    from radar_depth_amd import convdesc as cd
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    n, ci, co, k, s, h, w = cfg
    L = lib()
    g = torch.Generator().manual_seed(sum(cfg) + 1)
    upproj = k == 5
    if upproj:
        d = cd.upproj_fwd(n, h, w, ci, co)
        x = _bf(torch.randn(n, ci, h, w, generator=g))
        up = torch.zeros(n, ci, 2 * h, 2 * w)
        up[:, :, ::2, ::2] = x
        gy = _bf(torch.randn(n, co, 2 * h, 2 * w, generator=g))
        ref = torch.nn.grad.conv2d_weight(up.double(), (co, ci, 5, 5), gy.double(), 1, 2)
    else:
        d = cd.conv_fwd(n, h, w, ci, co, k, s, k // 2)
        x = _bf(torch.randn(n, ci, h, w, generator=g))
        gy = _bf(torch.randn(n, co, d.Ho, d.Wo, generator=g))
        ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, k, k), gy.double(), s, k // 2)
    assert L.rd_wgrad_bf16_supported(C.byref(d)) == 1
    L.rd_wgrad_bf16_workspace_floats.restype = C.c_int64
    slabs = torch.empty(int(L.rd_wgrad_bf16_workspace_floats(C.byref(d))), device="cuda")
    gw = torch.full((co, ci, k, k), float("nan"), device="cuda")
    xs, gys = _nhwc16(x), _nhwc16(gy)          # (named: a temporary would be freed -- and its block reused -- before the launch)
    check(L.rd_wgrad_bf16_t(BF16, C.byref(d), ptr(xs), ptr(gys), ptr(slabs), current_stream()), "wgrad_bf16_t")
    check(L.rd_wgrad_bf16_reduce(C.byref(d), ptr(slabs), ptr(gw), co, ci, k, k, 0, 0, current_stream()), "wgrad_bf16_reduce")
    torch.cuda.synchronize()
    err = ((gw.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err


def test_norm_kernels_bf16_storage():
    """bn_act / BatchNorm backward (reduce + apply, lone and joined forms) on bf16 tensors against fp64 torch of the same
    bf16 inputs: outputs within one bf16 ulp, per-channel sums (fp32) within 1e-5."""
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    g = torch.Generator().manual_seed(5)
    M, Cc = 3 * 17 * 23, 64
    x1 = _bf(torch.randn(M, Cc, generator=g))
    x2 = _bf(torch.randn(M, Cc, generator=g))
    dy = _bf(torch.randn(M, Cc, generator=g))
    s1, t1 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    s2, t2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g) * 0.3
    dev = lambda t, dt=torch.float32: t.to(dt).cuda().contiguous()
    X1, X2, DY = dev(x1, torch.bfloat16), dev(x2, torch.bfloat16), dev(dy, torch.bfloat16)
    S1, T1, S2, T2 = dev(s1), dev(t1), dev(s2), dev(t2)
    # forward join: y = relu(s1*x1+t1 + s2*x2+t2)
    Y = torch.empty(M, Cc, dtype=torch.bfloat16, device="cuda")
    check(L.rd_bn_act_t(BF16, ptr(X1), Cc, ptr(S1), ptr(T1), ptr(X2), Cc, ptr(S2), ptr(T2), ptr(Y), Cc, C.c_int64(M), Cc, 1, current_stream()), "bn_act_t")
    z = s1.double() * x1.double() + t1.double() + s2.double() * x2.double() + t2.double()
    torch.cuda.synchronize()
    assert _ulp_err(Y, z.clamp_min(0)) <= 1.01
    # backward of the lone form out = relu(s1*x1+t1): sums and dx
    mean = x1.mean(0)
    var = x1.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = torch.rand(Cc, generator=g) + 0.5
    sc = gamma * invstd
    sh = -mean * sc + 0.1
    tiles = L.rd_bn_bwd_tiles(C.c_int64(M), Cc)
    red = torch.zeros(tiles, 3, Cc, device="cuda")
    MEAN, INV, GAM, SC, SH = dev(mean), dev(invstd), dev(gamma), dev(sc), dev(sh)
    check(L.rd_bn_bwd_reduce_x_t(BF16, ptr(DY), Cc, ptr(X1), Cc, ptr(MEAN), ptr(SC), ptr(SH), None, 0, C.c_int64(M), Cc, 1, ptr(red), current_stream()),
          "bn_bwd_reduce_x_t")
    DX = torch.empty(M, Cc, dtype=torch.bfloat16, device="cuda")
    dg, db, coef = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), torch.zeros(3 * Cc, device="cuda")
    check(L.rd_bn_bwd_apply_x_t(BF16, ptr(DY), Cc, ptr(X1), Cc, ptr(red), tiles, ptr(GAM), ptr(MEAN), ptr(INV), ptr(SC), ptr(SH), 1, ptr(dg), ptr(db),
                                ptr(coef), ptr(DX), Cc, C.c_int64(M), Cc, current_stream()), "bn_bwd_apply_x_t")
    torch.cuda.synchronize()
    mask = ((SC.cpu().double() * x1.double() + SH.cpu().double()) > 0).double()       # the kernel's own fp32 coefficients
    gm = dy.double() * mask
    xc = x1.double() - mean.double()
    s_g, s_gx = gm.sum(0), (gm * xc).sum(0)
    assert ((db.double().cpu() - s_g).abs().max() / s_g.abs().max()).item() < 1e-5
    assert ((dg.double().cpu() - invstd.double() * s_gx).abs().max() / (invstd.double() * s_gx).abs().max()).item() < 1e-5
    dxr = gamma.double() * invstd.double() * (gm - s_g / M - xc * invstd.double() ** 2 * s_gx / M)
    assert _ulp_err(DX, dxr) <= 1.5


def _emulate_bf16_storage(model):
    import importlib.util
    spec = importlib.util.spec_from_file_location("bf16_emulation", os.path.join(HERE, "bf16_emulation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.emulate_bf16_storage(model)


def _pair(arch, h, w):
    from oracle import train as otrain
    from radar_depth_amd import main as hmain
    from radar_depth_amd.synthetic import procedural_fill_
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    mh, mo = hmain.create_model(args, [h, w]), otrain.create_model(args, [h, w])
    hm, hw_ = mh if isinstance(mh, tuple) else (mh, None)
    om, ow = mo if isinstance(mo, tuple) else (mo, None)
    procedural_fill_(hm)
    procedural_fill_(om)
    return args, hm.cuda().train(), hw_, om.train(), ow


@pytest.mark.parametrize("arch", ["resnet18_latefusion", pytest.param("resnet18_multistage_uncertainty_fixs", marks=pytest.mark.slow)])
def test_bf16_storage_train_step_vs_emulated_oracle(arch):
    """One training step with bf16 storage against the oracle with the same rounding points (tests/bf16_emulation.py).
    The quantised network is CHAOTIC at the level of individual pixels: tests/test_conditioning.py measures that a 1e-7 relative
    weight perturbation moves the emulated oracle's own output map by 6.6 % rms / 9.5 % max (1e-5 relative in fp32) -- a value
    that crosses a bf16 rounding boundary jumps by 2^-8 and batch-statistics BatchNorm over 48 values per channel amplifies it.
    Two correct implementations therefore agree on the map only to that floor; what is pinned tightly is what averages over
    pixels.  Stated tolerances (measured: latefusion / multistage): loss 2e-3 (6.8e-5 / 2.0e-4); gradient norm of EVERY parameter
    tensor within ~2x the oracle's own sensitivity floor (see the comment at the assertion); output map 0.15 max / 0.12 rms = 1.5x the self-sensitivity floor
    (8.1e-2 / 6.1e-2; stage 2 of the multistage net teacher-forced, see tests/test_gpu_configs.py); latefusion tail
    (decoder.layer4, conv3) gradients element-wise 0.15 norm-wise (5.6e-2)."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 97, 161
    args, hm, hw_, om, ow = _pair(arch, h, w)
    multistage = hw_ is not None
    assert _emulate_bf16_storage(om) == (104 if multistage else 52)
    x, t = make_batch(b, h, w, 300, ref_pixels=h * w)
    crit = otrain.make_criterion(args.arch)
    lo, po, ex = otrain.compute_loss(args.arch, om, crit, x, t, ow)
    lo.backward()
    init = [p.detach().clone() for p in hm.parameters()]
    ts = HipTrainStep(hm, b, h, w, lr=1.0, momentum=0.0, weight_decay=0.0, loss_weights=hw_, storage="bf16")     # update == gradient
    assert ts.plan.cat.t.dtype == torch.bfloat16
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    e_loss = abs(loss.item() - lo.item()) / abs(lo.item())
    ref_map = po.detach()
    if multistage:
        with torch.no_grad():
            p1h = ts.mp.p1.pred.detach().cpu()
            kept_o, _ = om.filter_layer(x[:, 3:4], p1h)
            ref_map = om.stage2(torch.cat((x[:, :3], kept_o, p1h), 1))
    e_out = ((pred.cpu() - ref_map).abs().max() / ref_map.abs().max()).item()
    e_rms = ((pred.cpu() - ref_map).norm() / ref_map.norm()).item()
    names = [n for n, _ in om.named_parameters()]
    go = [p.grad for p in om.parameters()]
    gg = [i0.cpu() - p.detach().cpu() for i0, p in zip(init, hm.parameters())]
    no = np.array([g_.double().norm().item() for g_ in go])
    ng = np.array([g_.double().norm().item() for g_ in gg])
    e_norm = np.abs(no - ng).max() / no.max()
    body = [i for i, n in enumerate(names) if not n.endswith(("conv1.weight", "conv1_depth.weight")) or ".layer" in n]   # all but the 7x7 stems
    e_body = np.abs(no[body] - ng[body]).max() / no.max()
    tail = [i for i, n in enumerate(names) if ("decoder.layer4" in n or "conv3" in n) and not n.startswith("stage1.")]
    e_tail = max((go[i] - gg[i]).norm().item() / max(go[i].norm().item(), 1e-20) for i in tail)
    print("bf16 storage %s: loss %.3e  out max %.3e rms %.3e  grad norms %.3e (%s)  tail elementwise %.3e"
          % (arch, e_loss, e_out, e_rms, e_norm, names[int(np.abs(no - ng).argmax())], e_tail))
    assert all(torch.isfinite(p).all() for p in hm.parameters())
    # gradient norms: tests/test_conditioning.py::test_bf16_storage_gradient_norm_floor measures how far the emulated oracle's OWN
    # norms move under a 1e-7 / +-1e-6 relative weight perturbation -- latefusion 1.2e-2 (stem weights; 3.8e-3 elsewhere),
    # multistage 1.1e-1 (stage-1 stem weights, reached through stage 2 and the radar filter; 3.4e-2 elsewhere).  Tile shapes and
    # summation orders of correct kernels move the HIP result inside that floor (measured over kernel revisions: latefusion
    # 8e-3 ... 1.9e-2, multistage 1.1e-2 ... 9.3e-2, always the stem weights), so the bounds are ~2x the floor.
    tol_all, tol_body = (0.2, 7e-2) if multistage else (4e-2, 1.5e-2)
    assert e_loss < 2e-3 and e_out < 0.15 and e_rms < 0.12 and e_norm < tol_all and e_body < tol_body, (e_norm, e_body)
    assert multistage or e_tail < 0.15


def test_bf16_storage_tracks_fp32_and_is_reproducible():
    """Three SGD steps: the losses stay within 1e-2 of the fp32 HIP step's; two identically initialised bf16-storage models stay
    bit-identical (deterministic reductions, no races between the role-split waves / streams)."""
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    b, h, w = 3, 97, 161
    runs = {}
    for tag, st in (("fp32", "fp32"), ("a", "bf16"), ("b", "bf16")):
        torch.manual_seed(0)
        m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
        procedural_fill_(m)
        m = m.cuda()
        ts = HipTrainStep(m, b, h, w, storage=st)
        losses = []
        for it in range(3):
            x, t = make_batch(b, h, w, 900 + it, ref_pixels=h * w)
            loss, _ = ts.step(x.cuda(), t.cuda())
            losses.append(loss.item())
        torch.cuda.synchronize()
        runs[tag] = (losses, [p.detach().clone() for p in m.parameters()])
    for a_, c_ in zip(runs["fp32"][0], runs["a"][0]):
        assert abs(a_ - c_) / abs(a_) < 1e-2, (runs["fp32"][0], runs["a"][0])
    assert runs["a"][0] == runs["b"][0] and runs["a"][0] != runs["fp32"][0]
    for p, q in zip(runs["a"][1], runs["b"][1]):
        assert torch.equal(p, q) and torch.isfinite(p).all()


@pytest.mark.parametrize("geom", [(1, 131, 77), (2, 228, 304), (1, 900, 1600)])
def test_bf16_storage_geometries(geom):
    """Ragged tiles / odd sizes / config 5's 900x1600 under bf16 storage: eval forward within the stated 3e-2 of the fp32 HIP
    forward (max-norm), one training step finite."""
    from radar_depth_amd.main import HipInference, HipTrainStep
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.synthetic import make_batch, procedural_fill_
    b, h, w = geom
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    m = m.cuda().eval()
    x, t = make_batch(b, h, w, 77, ref_pixels=min(h * w, 450 * 800))
    x, t = x.cuda(), t.cuda()
    ref = HipInference(m, b, h, w, use_graph=False)(x).clone()
    got = HipInference(m, b, h, w, use_graph=False, storage="bf16")(x).clone()
    torch.cuda.synchronize()
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    assert 0.0 < err <= 3e-2, err
    ts = HipTrainStep(m, b, h, w, storage="bf16")
    loss, pred = ts.step(x, t)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).all() and torch.isfinite(pred).all() and all(torch.isfinite(p).all() for p in m.parameters())


def test_config3_per_gpu_workload_b16_450x800_bf16_storage():
    """BASELINE configs[2]'s per-GPU workload exactly (resnet18_latefusion, b=16, 450x800) under bf16 storage: the training step's
    loss against the oracle with the plan's rounding points (2e-3), the forward map to the chaos floor of the quantised network
    (0.15 max / 0.12 rms, see test_bf16_storage_train_step_vs_emulated_oracle), everything finite after the update."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 16, 450, 800
    args, hm, hw_, om, ow = _pair("resnet18_latefusion", h, w)
    assert _emulate_bf16_storage(om) == 52
    x, t = make_batch(b, h, w, 1234)
    crit = otrain.make_criterion(args.arch)
    with torch.no_grad():
        lo, po, _ = otrain.compute_loss(args.arch, om, crit, x, t, ow)
    ts = HipTrainStep(hm, b, h, w, storage="bf16")
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    e_loss = abs(loss.item() - lo.item()) / abs(lo.item())
    e_max = ((pred.cpu() - po).abs().max() / po.abs().max()).item()
    e_rms = ((pred.cpu() - po).norm() / po.norm()).item()
    print("config 3 per-GPU workload, bf16 storage: loss %.3e  map max %.3e rms %.3e" % (e_loss, e_max, e_rms))
    assert e_loss < 2e-3 and e_max < 0.15 and e_rms < 0.12
    assert all(torch.isfinite(p).all() for p in hm.parameters())


@pytest.mark.parametrize("b", [pytest.param(2, marks=pytest.mark.slow), pytest.param(8, marks=pytest.mark.slow)])      # (900x1600 stays in the default run through test_gpu_model.py::test_large_geometry_900x1600 and test_gpu_bf16.py::test_bf16_large_geometry_900x1600)
def test_config5_geometry_multistage_900x1600_bf16_storage(b):
    """BASELINE configs[4]'s network and geometry (multistage_uncertainty_fixs, 900x1600) under bf16 storage, at b=2 and at the
    configuration's own per-GPU batch b=8 (what `bench.py --config 5` runs): the four loss terms of the fused step against the
    oracle with the plan's rounding points (2e-3); stage-1 map to the chaos floor; finite parameters after the update."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    h, w = 900, 1600
    args, hm, hw_, om, ow = _pair("resnet18_multistage_uncertainty_fixs", h, w)
    assert _emulate_bf16_storage(om) == 104
    x, t = make_batch(b, h, w, 4321)
    crit = otrain.make_criterion(args.arch)
    with torch.no_grad():
        lo, po, ex = otrain.compute_loss(args.arch, om, crit, x, t, ow)
    ts = HipTrainStep(hm, b, h, w, loss_weights=hw_, storage="bf16")
    loss, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    want4 = np.array([ex["d1"].item(), ex["d2"].item(), ex["smooth"].item(), lo.item()])
    got4 = ts.loss4.cpu().numpy()
    e_loss = np.abs(got4 - want4).max() / np.abs(want4).max()
    e1 = ((ts.mp.p1.pred.cpu() - ex["pred1"]).abs().max() / ex["pred1"].abs().max()).item()
    print("config 5 geometry, bf16 storage: losses %.3e  stage-1 map max %.3e" % (e_loss, e1))
    assert e_loss < 2e-3 and e1 < 0.15
    assert all(torch.isfinite(p).all() for p in hm.parameters())


@pytest.mark.slow
def test_bf16_storage_loss_trajectory_vs_fp32_oracle_450x800():
    """A bf16 check whose expectation does NOT pass through tests/bf16_emulation.py: three SGD steps of resnet18_latefusion at
    450x800 (b=2) under bf16 storage against the plain fp32 CPU oracle (pinned to the reference by the golden vectors) taking
    the same three steps with torch.optim.SGD.  Stated band: every step's loss within 1e-2 relative of the fp32 oracle's --
    the pixel-averaged quantity is well conditioned (bf16 storage rounds each stored tensor to 2^-9 relative; the masked-L1
    mean over ~6000 valid pixels averages that down), whereas individual pixels of the quantised network are chaotic
    (tests/test_conditioning.py).  A rounding point misplaced in BOTH the kernels and the emulation would pass the emulated
    tests and fail here."""
    from oracle import train as otrain
    from radar_depth_amd.main import HipTrainStep
    from radar_depth_amd.synthetic import make_batch
    b, h, w = 2, 450, 800
    args, hm, hw_, om, ow = _pair("resnet18_latefusion", h, w)
    crit = otrain.make_criterion(args.arch)
    opt = torch.optim.SGD(om.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    ts = HipTrainStep(hm, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, storage="bf16")
    got, want = [], []
    for it in range(3):
        x, t = make_batch(b, h, w, 5150 + it)
        lo, _, _ = otrain.train_step(args.arch, om, crit, opt, x, t, ow)
        loss, _ = ts.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        got.append(loss.item())
        want.append(lo.item())
    errs = [abs(g - w_) / abs(w_) for g, w_ in zip(got, want)]
    print("bf16 storage vs fp32 oracle, 3 steps at 450x800: losses", got, want, "rel", errs)
    assert max(errs) < 1e-2, (got, want)
    assert got != want                      # it is the bf16 path that ran
