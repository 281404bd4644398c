"""GPU parity of the generalised MFMA convolution (rd_gconv / rd_pack_weights) against torch CPU fp32
convolutions, through the C ABI.  Tolerance: 2e-5 relative to the output's max magnitude (fp32 MFMA is an
exact-fp32 fma chain; only the summation order differs from oneDNN)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _run_fwd(n, ci, co, k, s, p, h, w, seed=0):
    from radar_depth_amd import convdesc as cd, ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
    xg = ops.nchw_to_nhwc(x.cuda())
    wp = ops.pack_weights(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
    tiles = ops.gconv_stat_tiles(d)
    stat = torch.zeros(tiles, 2, co, device="cuda")
    ops.gconv(d, xg, wp, out, stat=stat)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, y) < 2e-5, (_rel(got, y), (n, ci, co, k, s, p, h, w))
    s_ = stat.sum(0).cpu().double()
    ref_s = y.double().sum((0, 2, 3))
    ref_q = (y.double() ** 2).sum((0, 2, 3))
    assert ((s_[0] - ref_s).abs().max() / ref_q.sqrt().max()).item() < 1e-4
    assert _rel(s_[1], ref_q) < 1e-4
    return x, wt, y


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 3, 1, 1, 113, 200),   # layer1
    (2, 64, 128, 3, 2, 1, 113, 200),  # layer2.0.conv1
    (2, 128, 128, 3, 1, 1, 57, 100),
    (2, 64, 128, 1, 2, 0, 113, 200),  # downsample
    (2, 256, 256, 3, 1, 1, 29, 50),
    (2, 512, 512, 3, 1, 1, 15, 25),   # layer4
    (2, 640, 512, 1, 1, 0, 15, 25),   # conv_fusion
    (2, 16, 16, 3, 1, 1, 113, 200),   # depth layer1
    (2, 16, 32, 3, 2, 1, 113, 200),
    (2, 32, 32, 3, 1, 1, 120, 200),   # decoder.layer3 conv2
    (1, 16, 16, 3, 1, 1, 240, 400),   # decoder.layer4 conv2
    (3, 32, 48, 3, 1, 1, 9, 7),       # tiny / ragged
    (16, 512, 512, 3, 1, 1, 15, 25),  # B=16: the planner splits the input channels (split-K + combine kernel)
    (16, 512, 256, 1, 1, 0, 15, 25),
    (16, 128, 128, 3, 1, 1, 15, 25),
    # ragged geometry for the padded / swizzled LDS layouts and the quad-interleaved weight operand: output channels that do not
    # fill a 32/64 block, 1-pixel-wide and single-pixel images, stride 2 on odd sizes, 48/80/96-channel reductions
    (2, 48, 80, 3, 1, 1, 31, 17),
    (1, 96, 36, 3, 2, 1, 33, 45),
    (3, 64, 64, 3, 1, 1, 1, 1),
    (2, 32, 16, 3, 1, 1, 40, 1),
    (2, 80, 48, 1, 1, 0, 19, 23),
    (4, 16, 64, 1, 2, 0, 7, 5),
    # csrc/conv16.hip (16 -> 16 channels on the 16x16x4 MFMA): partial 16x16 tiles on both edges, images smaller than a tile
    (3, 16, 16, 3, 1, 1, 37, 45),
    (2, 16, 16, 3, 1, 1, 16, 16),
    (5, 16, 16, 3, 1, 1, 1, 1),
    (2, 16, 16, 3, 1, 1, 3, 50),
])
def test_gconv_forward(cfg):
    _run_fwd(*cfg)


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 3, 1, 1, 113, 200),
    (2, 64, 128, 3, 2, 1, 113, 200),
    (2, 64, 128, 1, 2, 0, 113, 200),
    (2, 256, 512, 3, 2, 1, 29, 50),
    (2, 16, 32, 3, 2, 1, 57, 101),
    (16, 512, 512, 3, 1, 1, 15, 25),  # split-K dgrad with the residual addend applied by the combine kernel
    (2, 48, 80, 3, 1, 1, 31, 17),
    (1, 96, 48, 3, 2, 1, 33, 45),
    (3, 64, 64, 3, 1, 1, 1, 1),
    (2, 16, 16, 3, 1, 1, 113, 200),   # csrc/conv16.hip: input gradient (flipped taps, transposed operand) + residual addend
    (3, 16, 16, 3, 1, 1, 37, 45),
    (2, 16, 16, 3, 1, 1, 1, 7),
])
def test_gconv_dgrad(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * co)) ** 0.5
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
    wp = ops.pack_weights(wt.cuda(), transpose=True)
    dy = ops.nchw_to_nhwc(gy.cuda())
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    if zero_fill:
        ops.fill(dx, 0.0)
    add = torch.randn(n, h, w, ci, generator=g)
    ops.gconv(d, dy, wp, dx, addend=add.cuda() if not zero_fill else None, ld_add=ci)
    torch.cuda.synchronize()
    want = x.grad + (add.permute(0, 3, 1, 2) if not zero_fill else 0)
    got = dx.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, want) < 2e-5


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (32, 13, 9)])
def test_gconv_upproj(c, h, w):
    from radar_depth_amd import convdesc as cd, ops
    n = 2
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    w_up = torch.randn(c // 2, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    w_bt = torch.randn(c // 2, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    u = torch.zeros(n, c, 2 * h, 2 * w)
    xs = x.detach()
    u[:, :, ::2, ::2] = xs
    u.requires_grad_(True)
    wcat = torch.cat((w_up, w_bt), 0)
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    gx_want = u.grad[:, :, ::2, ::2]
    d = cd.upproj_fwd(n, h, w, c, c)
    wp = torch.zeros(25, c, c, device="cuda")
    ops.pack_weights(w_up.cuda(), out=wp, ldc=c, off=0)
    ops.pack_weights(w_bt.cuda(), out=wp, ldc=c, off=c // 2)
    out = torch.full((n, 2 * h, 2 * w, c), float("nan"), device="cuda")
    stat = torch.zeros(ops.gconv_stat_tiles(d), 2, c, device="cuda")
    ops.gconv(d, ops.nchw_to_nhwc(xs.cuda()), wp, out, stat=stat)
    dd = cd.upproj_dgrad(n, h, w, c, c)
    wd = torch.zeros(25, c, c, device="cuda")
    ops.pack_weights(w_up.cuda(), transpose=True, out=wd, ldc=c, off=0, rows_total=c)
    ops.pack_weights(w_bt.cuda(), transpose=True, out=wd, ldc=c, off=c // 2, rows_total=c)
    dx = torch.full((n, h, w, c), float("nan"), device="cuda")
    ops.gconv(dd, ops.nchw_to_nhwc(gy.cuda()), wd, dx)
    torch.cuda.synchronize()
    got = out.permute(0, 3, 1, 2).cpu()
    assert not torch.isnan(got).any()
    assert _rel(got, y.detach()) < 2e-5
    assert _rel(stat.sum(0)[0].cpu().double(), y.detach().double().sum((0, 2, 3))) < 1e-3
    gotx = dx.permute(0, 3, 1, 2).cpu()
    assert _rel(gotx, gx_want) < 2e-5


def test_gconv_launches_are_bitwise_reproducible():
    """Race regression (round 2): the single-block tile with 16-channel chunks in the pipelined loop -- the plan the heuristic picks
    for a 64 -> 64 channel 3x3 layer at b=2, 113x200 -- produced two wrong output pixels per launch in 1.7 % of the launches: hipcc
    had dropped the lgkmcnt(0) wait in front of the loop-header barrier, so a wave could read patch pixels another wave had stored
    but not committed (csrc/common.h rd_sync / glds_wait).  600 launches on the same inputs must agree bit for bit, output and
    BatchNorm partial sums; tools/stress_plans.py does the same for every candidate plan of a descriptor."""
    import ctypes as C
    from radar_depth_amd import convdesc as cd
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    L = lib()
    for n, h, w, ci, co in ((2, 113, 200, 64, 64), (2, 57, 100, 128, 128)):
        d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(n * h * w * ci, device="cuda", generator=g)
        wp = torch.randn(9 * ci * co, device="cuda", generator=g)
        out = torch.zeros(n * h * w * co, device="cuda")
        stat = torch.zeros(L.rd_gconv_stat_tiles_ws(C.byref(d)) * 2 * co, device="cuda")
        L.rd_gconv_workspace_floats.restype = C.c_int64
        nws = int(L.rd_gconv_workspace_floats(C.byref(d)))
        ws = torch.empty(max(nws, 1), device="cuda")
        ref = None
        for it in range(600):
            out.fill_(float("nan"))
            check(L.rd_gconv_ws(C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, ptr(stat), ptr(ws) if nws else None, current_stream()), "gconv")
            if it % 50 == 0 or it == 599:
                torch.cuda.synchronize()
            cur = (out.clone(), stat.clone())
            if ref is None:
                ref = cur
            else:
                assert torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1]), (n, h, w, ci, co, it)


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 113, 200, 1),      # layer1 conv2 (ReLU)
    (2, 128, 128, 57, 100, 1),
    (4, 32, 32, 57, 100, 1),       # depth layer2
    (2, 48, 80, 31, 17, 2),        # ragged channels / tiles, LeakyReLU(0.2)
    (3, 64, 32, 9, 7, 1),
    (2, 256, 256, 29, 50, 1),
])
def test_gconv_bnbwd_epilogue(cfg):
    """rd_gconv_bnbwd: the input gradient dy of conv(act(s*x + t)) with the BatchNorm-backward sums of that BatchNorm taken in the
    epilogue -- sum g and sum g*(x - mean), g = dy * act'(s*x + t) -- against the plain dgrad and float64 sums (5e-5 of each
    vector's largest magnitude: summation order only).  x lives in a WIDER buffer (channel slice), as in the UpProj modules."""
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import lib
    import ctypes as C
    n, ci, co, h, w, act = cfg
    g = torch.Generator().manual_seed(11)
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * co)) ** 0.5
    gy = torch.randn(n, co, h, w, generator=g)
    d, zero_fill = cd.conv_dgrad(n, h, w, ci, co, 3, 1, 1)
    assert not zero_fill
    if lib().rd_gconv_bnbwd_supported(C.byref(d)) != 1:
        pytest.skip("this descriptor's plan splits the reduction")
    wp = ops.pack_weights(wt.cuda(), transpose=True)
    dy_in = ops.nchw_to_nhwc(gy.cuda())
    dx_ref = torch.empty(n, h, w, ci, device="cuda")
    ops.gconv(d, dy_in, wp, dx_ref)
    ld = ci + 8
    xbuf = torch.randn(n, h, w, ld, generator=g).cuda()
    x = xbuf[..., 4:4 + ci]
    scale = (torch.rand(ci, generator=g) + 0.5).cuda() * (torch.randint(0, 2, (ci,), generator=g).cuda() * 2 - 1)
    shift = torch.randn(ci, generator=g).cuda() * 0.3
    mean = torch.randn(ci, generator=g).cuda() * 0.2
    tiles = ops.gconv_stat_tiles(d)
    red = torch.full((tiles, 3, ci), float("nan"), device="cuda")
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    xs = xbuf.view(-1)[4:]
    ops.gconv_bnbwd(d, dy_in, wp, dx, xs, ld, mean, scale, shift, act, red)
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)                       # the convolution itself is untouched
    z = (scale * x + shift).double()
    slope = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.0 if act == 1 else 0.2))
    gg = dx_ref.double() * slope
    want0 = gg.sum((0, 1, 2))
    want1 = (gg * (x.double() - mean.double())).sum((0, 1, 2))
    got = red[:, :2].double().sum(0)
    assert ((got[0] - want0).abs().max() / want0.abs().max()).item() < 5e-5
    assert ((got[1] - want1).abs().max() / want1.abs().max()).item() < 5e-5
