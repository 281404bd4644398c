"""GPU parity of the 7x7 / stride-2 stem forward kernels (rd_stem_fwd: csrc/stem.hip for the RGB stem, csrc/stem16.hip for the 16-channel
depth stem with one or two input planes) against torch CPU fp32 conv2d, through the C ABI, incl. the BatchNorm partial sums.
Tolerance 2e-5 of the output's max magnitude (exact-fp32 MFMA chain; only the summation order differs)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [
    (2, 3, 64, 97, 161),      # RGB stem, ragged tiles
    (2, 1, 16, 97, 161),      # depth stem
    (2, 2, 16, 97, 161),      # stage 2 of the multistage net: two depth planes
    (3, 1, 16, 450, 800),     # bench geometry
    (1, 1, 16, 15, 63),       # a single ragged tile row
    (2, 2, 16, 64, 64),
])
def test_stem_forward(cfg):
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin + 1, h, w, generator=g)          # the stem reads planes 1.. of a wider NCHW tensor (strided images)
    wt = torch.randn(cout, cin, 7, 7, generator=g) * 0.1
    y = F.conv2d(x[:, 1:], wt, stride=2, padding=3)
    xg = x.cuda()
    hw = h * w
    planes = (C.c_void_p * 3)(*[xg.data_ptr() + 4 * hw * (1 + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[(cin + 1) * hw if c < cin else 0 for c in range(3)])
    wp = wt.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous().cuda()
    ho, wo = y.shape[2], y.shape[3]
    out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
    tiles = L.rd_stem_stat_tiles(n, h, w)
    stat = torch.zeros(tiles, 2, cout, device="cuda")
    check(L.rd_stem_fwd(planes, strides, cin, n, h, w, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), "rd_stem_fwd")
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert ((got - y).abs().max() / y.abs().max()).item() < 2e-5, cfg
    s_ = stat.sum(0).cpu().double()
    ref_s, ref_q = y.double().sum((0, 2, 3)), (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - ref_s).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert ((s_[1] - ref_q).abs().max() / ref_q.max()).item() < 1e-4


@pytest.mark.parametrize("cfg", [
    (2, 3, 64, 97, 161),      # RGB stem, ragged tiles
    (16, 3, 64, 450, 800),    # BASELINE config 2's own stem launch
    (2, 1, 16, 97, 161),      # (the kernel also takes the depth stem's shapes: one 16-wide channel tile)
    (2, 2, 32, 64, 64),
    (1, 3, 64, 15, 63),       # a single ragged tile row
])
def test_stem_forward_split(cfg):
    """rd_stem_fwd_split (csrc/stem_bf16.hip, NP = 3: input and weights split into three bf16 pieces while staged, six MFMAs per product,
    fp32 accumulation) against an fp64 convolution: as close as the fp32-MFMA stem (2e-6 of the output's max magnitude here; the fp32
    kernel is held to 2e-5 against torch CPU fp32), same partial-sum layout, inputs spanning 2^-20 .. 2^20 in magnitude."""
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w = cfg
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, cin + 1, h, w, generator=g)
    x[:, 1:, : h // 3] *= 2.0 ** 20                         # dynamic range: the pieces of large and of tiny values
    x[:, 1:, 2 * h // 3:] *= 2.0 ** -20
    wt = torch.randn(cout, cin, 7, 7, generator=g) * 0.1
    xg, wg = x.cuda(), wt.cuda()
    y = F.conv2d(xg[:, 1:].double(), wg.double(), stride=2, padding=3)
    hw = h * w
    planes = (C.c_void_p * 3)(*[xg.data_ptr() + 4 * hw * (1 + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[(cin + 1) * hw if c < cin else 0 for c in range(3)])
    wp = wt.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous().cuda()
    ho, wo = y.shape[2], y.shape[3]
    tiles = L.rd_stem_stat_tiles(n, h, w)
    res = {}
    for name, fn in (("split", L.rd_stem_fwd_split), ("fp32", L.rd_stem_fwd)):
        out = torch.full((n, ho, wo, cout), float("nan"), device="cuda")
        stat = torch.zeros(tiles, 2, cout, device="cuda")
        check(fn(planes, strides, cin, n, h, w, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), name)
        torch.cuda.synchronize()
        got = out.permute(0, 3, 1, 2).double()
        assert not torch.isnan(got).any()
        # per third of the image (each has its own magnitude): error relative to that third's largest output
        errs = []
        for lo, hi in ((0, ho // 3 - 2), (ho // 3 + 2, 2 * ho // 3 - 2), (2 * ho // 3 + 2, ho)):
            if hi > lo:
                errs.append(((got[:, :, lo:hi] - y[:, :, lo:hi]).abs().max() / y[:, :, lo:hi].abs().max()).item())
        res[name] = (max(errs), stat)
    assert res["split"][0] < 2e-6, (cfg, res["split"][0], res["fp32"][0])
    assert res["split"][0] < 2.0 * res["fp32"][0] + 1e-7
    s_, f_ = res["split"][1].sum(0).double(), res["fp32"][1].sum(0).double()
    assert ((s_ - f_).abs().max() / f_.abs().max()).item() < 1e-5


@pytest.mark.parametrize("cfg", [(2, 2, 16, 97, 161, 1), (8, 2, 16, 450, 800, 1), (1, 2, 16, 15, 63, 0), (2, 2, 16, 64, 64, 1), (2, 3, 64, 33, 47, 2)])
def test_stem_dgrad_channel(cfg):
    """rd_stem_dgrad_channel (the gradient w.r.t. ONE input plane of the 7x7 / stride-2 stem: stage 2's dense-depth channel,
    multistage_model.py:75) against torch autograd: the LDS-tiled 16-channel form (parity-class waves) and the generic kernel (Cout = 64)."""
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w, ci = cfg
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, 7, 7, generator=g, dtype=torch.float64) * 0.1
    y = F.conv2d(x, wt, stride=2, padding=3)
    go = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(go)
    ref = x.grad[:, ci]
    dout = go.permute(0, 2, 3, 1).contiguous().float().cuda()
    wp = wt.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous().float().cuda()
    dx = torch.full((n, h, w), float("nan"), device="cuda")
    check(L.rd_stem_dgrad_channel(ptr(dout), ptr(wp), n, h, w, cin, ci, cout, ptr(dx), current_stream()), "rd_stem_dgrad_channel")
    torch.cuda.synchronize()
    assert not torch.isnan(dx).any()
    assert ((dx.cpu().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6, cfg


@pytest.mark.parametrize("cfg", [
    (2, 3, 64, 97, 161, 0),       # RGB stem, ragged tiles in both directions
    (2, 1, 16, 97, 161, 0),       # depth stem
    (2, 2, 16, 64, 64, 0),        # stage 2 of the multistage net: two depth planes
    (16, 3, 64, 450, 800, 0),     # BASELINE config 2's own launch (one workgroup per CU, ~46 tiles each)
    (1, 3, 64, 15, 63, 0),        # a single ragged tile row, fewer tiles than CUs
    (3, 3, 64, 9, 11, 0),         # tiles smaller than the 4 x 32 block
    (2, 3, 64, 97, 161, 1),       # bf16 storage: dout is its own single piece
    (2, 1, 16, 97, 161, 1),
    (8, 3, 64, 450, 800, 1),
])
def test_stem_wgrad_split(cfg):
    """rd_stem_wgrad_split_t (csrc/stem_wgrad_split.hip: three-piece operands on v_mfma_f32_32x32x16_bf16, fp32 accumulation) against an
    fp64 weight gradient: as close as the fp32-MFMA kernel rd_stem_wgrad_t (bar: 1.5x its error + a floor), inputs spanning 2^-12 .. 2^12 in
    magnitude, every element of the gradient written (NaN-filled target and workspace)."""
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w, dt = cfg
    assert L.rd_stem_wgrad_split_supported(cin, cout) == 1
    tdt = torch.bfloat16 if dt else torch.float32
    gen = torch.Generator().manual_seed(17)
    mag = torch.exp2(torch.randint(-12, 13, (n, cin + 1, 1, 1), generator=gen).float())
    x = (torch.randn(n, cin + 1, h, w, generator=gen) * mag).cuda()       # the stem reads planes 1.. of a wider NCHW tensor (strided images)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    dout = (torch.randn(n, ho, wo, cout, generator=gen) * torch.exp2(torch.randint(-6, 7, (1, 1, 1, cout), generator=gen).float())).to(tdt).cuda()
    hw = h * w
    planes = (C.c_void_p * 3)(*[x.data_ptr() + 4 * hw * (1 + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[(cin + 1) * hw if c < cin else 0 for c in range(3)])
    L.rd_stem_wgrad_workspace_floats.restype = C.c_int64
    nws = int(L.rd_stem_wgrad_workspace_floats(n, h, w, cin, cout))
    want = torch.nn.grad.conv2d_weight(x[:, 1:].double(), (cout, cin, 7, 7), dout.double().permute(0, 3, 1, 2).contiguous(), stride=2, padding=3).cpu()
    err = {}
    for name, fn in (("fp32", L.rd_stem_wgrad_t), ("split", L.rd_stem_wgrad_split_t)):
        ws = torch.full((nws,), float("nan"), device="cuda")
        gw = torch.full((cout, cin, 7, 7), float("nan"), device="cuda")
        check(fn(dt, planes, strides, cin, n, h, w, ptr(dout), cout, ptr(gw), ptr(ws), current_stream()), name)
        torch.cuda.synchronize()
        got = gw.cpu().double()
        assert not torch.isnan(got).any(), (cfg, name)
        err[name] = ((got - want).abs().max() / want.abs().max()).item()
    print("stem wgrad %s: fp32-MFMA %.2e, split %.2e of the gradient's max" % (cfg, err["fp32"], err["split"]))
    assert err["split"] < 1.5 * err["fp32"] + 2e-7, (cfg, err)
    assert err["split"] < 2e-5, (cfg, err)


@pytest.mark.parametrize("cfg", [
    (2, 3, 64, 97, 161, 0),       # RGB stem, ragged tiles
    (2, 1, 16, 97, 161, 0),       # depth stem (one 16-channel half of a 32-channel tile)
    (4, 3, 64, 225, 400, 0),      # several tiles per workgroup
    (1, 3, 64, 9, 11, 0),
    (2, 3, 64, 97, 161, 1),       # bf16 storage: the operand is the bf16-rounded value the separate pass would have stored
    (2, 1, 16, 97, 161, 1),
])
def test_stem_wgrad_split_bn_same_bits_as_two_passes(cfg):
    """rd_stem_wgrad_split_bn_t (the stem BatchNorm's backward apply pass folded into the staging waves of the split weight gradient) against
    rd_bn_bwd_apply_t + rd_stem_wgrad_split_t on the same tensors: bit-identical weight gradient, dgamma, dbeta (same expression, same
    order of operations, same pieces)."""
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    n, cin, cout, h, w, dt = cfg
    tdt = torch.bfloat16 if dt else torch.float32
    gen = torch.Generator().manual_seed(11)
    x_in = torch.randn(n, cin, h, w, generator=gen).cuda()
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    m = n * ho * wo
    raw = (torch.randn(n, ho, wo, cout, generator=gen) * 1.5 + 0.3).to(tdt).cuda()
    g = (torch.randn(n, ho, wo, cout, generator=gen) * (torch.rand(n, ho, wo, cout, generator=gen) < 0.3)).to(tdt).cuda()   # pooled gradients are sparse
    gamma = (torch.rand(cout, generator=gen) + 0.5).cuda()
    x32 = raw.float()
    mean = x32.mean((0, 1, 2))
    invstd = 1.0 / torch.sqrt(x32.var((0, 1, 2), unbiased=False) + 1e-5)
    tiles = L.rd_bn_bwd_tiles(C.c_int64(m), cout)
    red = torch.zeros(tiles, 3, cout, device="cuda")
    check(L.rd_bn_bwd_reduce_t(dt, ptr(g), cout, None, 0, ptr(raw), cout, ptr(mean), None, 0, None, None, 0, C.c_int64(m), cout, 0, ptr(red),
                               current_stream()), "rd_bn_bwd_reduce_t")
    hw = h * w
    planes = (C.c_void_p * 3)(*[x_in.data_ptr() + 4 * hw * c if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[cin * hw if c < cin else 0 for c in range(3)])
    L.rd_stem_wgrad_workspace_floats.restype = C.c_int64
    nws = L.rd_stem_wgrad_workspace_floats(n, h, w, cin, cout)
    res = []
    for fused in (0, 1):
        ws = torch.full((int(nws),), float("nan"), device="cuda")
        dg, db = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
        coef = torch.zeros(3 * cout, device="cuda")
        gw = torch.full((cout, cin, 7, 7), float("nan"), device="cuda")
        if fused:
            check(L.rd_stem_wgrad_split_bn_t(dt, planes, strides, cin, n, h, w, ptr(g), ptr(raw), ptr(red), tiles, ptr(gamma), ptr(mean), ptr(invstd),
                                             ptr(dg), ptr(db), ptr(coef), cout, ptr(gw), ptr(ws), current_stream()), "rd_stem_wgrad_split_bn_t")
        else:
            dx = torch.empty_like(raw)
            check(L.rd_bn_bwd_apply_t(dt, ptr(g), cout, ptr(raw), cout, ptr(red), tiles, 1, ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db),
                                      ptr(coef), ptr(dx), cout, C.c_int64(m), cout, current_stream()), "rd_bn_bwd_apply_t")
            check(L.rd_stem_wgrad_split_t(dt, planes, strides, cin, n, h, w, ptr(dx), cout, ptr(gw), ptr(ws), current_stream()), "rd_stem_wgrad_split_t")
        torch.cuda.synchronize()
        res.append((gw.cpu(), dg.cpu(), db.cpu()))
    for a_, b_ in zip(res[0], res[1]):
        assert not torch.isnan(b_).any()
        assert torch.equal(a_, b_), cfg
