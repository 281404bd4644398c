"""GPU parity of rd_wgrad_split (csrc/wgrad_split.hip: the fp32 weight gradient rebuilt from six bf16 MFMAs per product over three-piece
operands, fragments by ds_read_b64_tr_b16 from pixel-major LDS images) against torch CPU autograd at the fp32 kernel's tolerance
(tests/test_gpu_wgrad.py: 5e-5 of the gradient's max magnitude), and against an fp64 gradient side by side with rd_wgrad."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 113, 200),      # layer1
    (2, 128, 128, 57, 100),
    (2, 256, 256, 29, 50),
    (2, 512, 512, 15, 25),      # layer4
    (16, 64, 64, 60, 100),      # many tiles per split
    (3, 96, 160, 31, 51),       # partial 64-channel blocks on both sides, odd rows, ragged last column tile
    (1, 64, 64, 1, 1),          # a single pixel
    (2, 64, 72, 7, 33),         # one column past a tile boundary
    (5, 128, 64, 24, 3),
    (2, 72, 64, 2, 32),         # exactly one tile
])
def test_wgrad_split(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = cfg
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g, requires_grad=True)
    y = F.conv2d(x, wt, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    assert ops.wgrad_split_supported(d)
    slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    ops.wgrad_split(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs)
    grad = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    ops.wgrad_split_reduce(d, slabs, grad)
    torch.cuda.synchronize()
    assert not torch.isnan(grad).any()
    assert _rel(grad.cpu(), wt.grad) < 5e-5, (cfg, _rel(grad.cpu(), wt.grad))
    ops.wgrad_split_reduce(d, slabs, grad, accumulate=True)
    torch.cuda.synchronize()
    assert _rel(grad.cpu(), 2 * wt.grad) < 5e-5


def test_wgrad_split_unsupported_shapes_say_so():
    from radar_depth_amd import convdesc as cd, ops
    assert not ops.wgrad_split_supported(cd.conv_fwd(2, 57, 100, 64, 128, 3, 2, 1))      # stride 2
    assert not ops.wgrad_split_supported(cd.conv_fwd(2, 15, 25, 640, 512, 1, 1, 0))      # 1x1
    assert not ops.wgrad_split_supported(cd.conv_fwd(2, 57, 100, 32, 32, 3, 1, 1))       # < 64 channels
    assert not ops.wgrad_split_supported(cd.upproj_fwd(2, 15, 25, 32, 32))


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (128, 25, 51), (64, 5, 3), (96, 13, 33)])
def test_wgrad_split_upproj(c, h, w):
    """UpProj 5x5 (four parity phases of 9 / 6 / 6 / 4 taps against the output gradient sampled at stride 2, one launch each, 25 slabs),
    reduced into the two branch weights (column ranges of one slab set)."""
    from radar_depth_amd import convdesc as cd, ops
    n = 2
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, c, h, w, generator=g)
    wcat = torch.randn(c, c, 5, 5, generator=g, requires_grad=True)
    u = torch.zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.upproj_fwd(n, h, w, c, c)
    assert ops.wgrad_split_supported(d)
    slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    ops.wgrad_split(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs)
    g_up = torch.full((c // 2, c, 5, 5), float("nan"), device="cuda")
    g_bt = torch.full((c // 2, c, 5, 5), float("nan"), device="cuda")
    ops.wgrad_split_reduce(d, slabs, g_up, co_off=0)
    ops.wgrad_split_reduce(d, slabs, g_bt, co_off=c // 2)
    torch.cuda.synchronize()
    assert _rel(g_up.cpu(), wcat.grad[:c // 2]) < 5e-5
    assert _rel(g_bt.cpu(), wcat.grad[c // 2:]) < 5e-5


@pytest.mark.parametrize("cfg", [(4, 64, 64, 113, 200), (4, 512, 512, 15, 25), (4, 256, 256, 29, 50)])
def test_wgrad_split_is_as_close_to_fp64_as_the_fp32_mfma(cfg):
    """Error against an fp64 weight gradient of the same fp32 inputs, next to the fp32-MFMA kernel's and torch's CPU fp32 gradient."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = cfg
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, ci, h, w, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(gy.double())
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    xg, gg = ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda())
    s1 = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    s2 = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    g1 = torch.empty(co, ci, 3, 3, device="cuda")
    g2 = torch.empty(co, ci, 3, 3, device="cuda")
    ops.wgrad(d, xg, gg, s1)
    ops.wgrad_reduce(d, s1, g1)
    ops.wgrad_split(d, xg, gg, s2)
    ops.wgrad_split_reduce(d, s2, g2)
    torch.cuda.synchronize()
    e1 = g1.cpu().double() - wt.grad
    e2 = g2.cpu().double() - wt.grad
    # the same gradient from torch's CPU fp32 convolution (oneDNN): a third fp32 evaluation, for scale
    w32 = torch.zeros(co, ci, 3, 3, requires_grad=True)
    F.conv2d(x, w32, padding=1).backward(gy)
    e3 = w32.grad.double() - wt.grad
    scale = wt.grad.abs().max()
    print("fp32 MFMA: max %.3e rms %.3e | split: max %.3e rms %.3e | torch CPU fp32: max %.3e rms %.3e | gradient max %.3e"
          % (e1.abs().max(), e1.pow(2).mean().sqrt(), e2.abs().max(), e2.pow(2).mean().sqrt(), e3.abs().max(), e3.pow(2).mean().sqrt(), scale))
    # a long reduction (up to 360 k pixels per element): the dropped piece products (< 2^-24 of each product, random sign) add up next
    # to the accumulator roundings, so the split kernel sits a little above the fp32 MFMA kernel here -- by a factor below 2, and two
    # orders of magnitude inside the kernel tolerance (5e-5 of the gradient's max)
    assert e2.abs().max() <= 2.0 * max(e1.abs().max(), e3.abs().max())
    assert e2.pow(2).mean().sqrt() <= 2.0 * max(e1.pow(2).mean().sqrt(), e3.pow(2).mean().sqrt())
    assert e2.abs().max() < 5e-6 * scale


@pytest.mark.parametrize("ex,ey", [(60, -60), (-60, 60), (-30, -30), (-100, 90)])
def test_wgrad_split_dynamic_range(ex, ey):
    """Both operands of the weight gradient are split on the fly: scaled by 2^+-60 (x down to 2^-100) against an fp64 gradient at the
    kernel's own 5e-5 bar, side by side with rd_wgrad."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 64, 128, 29, 50
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, ci, h, w, generator=g) * 2.0 ** ex
    gy = torch.randn(n, co, h, w, generator=g) * 2.0 ** ey
    wt = torch.zeros(co, ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(gy.double())
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    xg, yg = ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda())
    res = []
    for split in (True, False):
        slabs = torch.empty(ops.wgrad_split_workspace_floats(d) if split else ops.wgrad_workspace_floats(d), device="cuda")
        grad = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
        if split:
            ops.wgrad_split(d, xg, yg, slabs)
            ops.wgrad_split_reduce(d, slabs, grad)
        else:
            ops.wgrad(d, xg, yg, slabs)
            ops.wgrad_reduce(d, slabs, grad)
        torch.cuda.synchronize()
        res.append(_rel(grad.cpu().double(), wt.grad))
    print("scale 2^%d x 2^%d: split %.3e, fp32 MFMA %.3e" % (ex, ey, res[0], res[1]))
    assert res[0] < 5e-5 and res[0] <= 2.0 * res[1] + 1e-7


def test_wgrad_split_non_finite():
    """A NaN / inf in either operand makes the taps whose reduction touches it non-finite in both kernels; weight-gradient elements of
    channels the poisoned values do not belong to stay bit-identical to the clean run."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = 2, 64, 64, 20, 33
    g = torch.Generator().manual_seed(14)
    x = torch.randn(n, ci, h, w, generator=g)
    gy = torch.randn(n, co, h, w, generator=g)
    xb, gb = x.clone(), gy.clone()
    xb[0, 5, 3, 4] = float("inf")
    gb[1, 9, 10, 11] = float("nan")
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)

    def run(xx, yy):
        slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
        grad = torch.empty(co, ci, 3, 3, device="cuda")
        ops.wgrad_split(d, ops.nchw_to_nhwc(xx.cuda()), ops.nchw_to_nhwc(yy.cuda()), slabs)
        ops.wgrad_split_reduce(d, slabs, grad)
        torch.cuda.synchronize()
        return grad.cpu()
    clean, bad = run(x, gy), run(xb, gb)
    assert torch.isfinite(clean).all()
    hit = torch.zeros(co, ci, dtype=torch.bool)
    hit[:, 5] = True
    hit[9, :] = True
    assert not torch.isfinite(bad[hit]).any()
    assert torch.equal(bad[~hit], clean[~hit])


# ------------------------------------------------------------------------------------------------ pre-split operands
@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 113, 200),      # layer1
    (2, 128, 128, 57, 100),
    (2, 256, 256, 29, 50),
    (2, 512, 512, 15, 25),      # layer4
    (16, 64, 64, 60, 100),      # many tiles per split
    (3, 96, 160, 31, 51),       # partial 64-channel blocks on both sides, odd rows, ragged last column tile
    (1, 64, 64, 1, 1),          # a single pixel
    (2, 64, 80, 7, 33),         # one column past a tile boundary
    (5, 128, 64, 24, 3),
    (2, 80, 64, 2, 32),         # exactly one tile
])
def test_wgrad_split_pre(cfg):
    """rd_wgrad_split_pre (both operands split by their producers, staging = global_load_lds only) at rd_wgrad_split's own bar."""
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, h, w = cfg
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g, requires_grad=True)
    y = F.conv2d(x, wt, padding=1)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    assert ops.wgrad_split_pre_supported(d)
    slabs = torch.full((ops.wgrad_split_workspace_floats(d),), float("nan"), device="cuda")
    xp, yp = ops.split_pieces(ops.nchw_to_nhwc(x.cuda())), ops.split_pieces(ops.nchw_to_nhwc(gy.cuda()))
    ops.wgrad_split_pre(d, xp, yp, slabs)
    grad = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    ops.wgrad_split_reduce(d, slabs, grad)
    torch.cuda.synchronize()
    assert not torch.isnan(grad).any()
    assert _rel(grad.cpu(), wt.grad) < 5e-5, (cfg, _rel(grad.cpu(), wt.grad))
    # the same products in the same order as the on-the-fly split: bit-identical slabs
    slabs2 = torch.empty_like(slabs)
    ops.wgrad_split(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs2)
    grad2 = torch.empty_like(grad)
    ops.wgrad_split_reduce(d, slabs2, grad2)
    torch.cuda.synchronize()
    assert torch.equal(grad, grad2)


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (128, 25, 51), (64, 5, 3), (96, 13, 33)])
def test_wgrad_split_pre_upproj(c, h, w):
    """UpProj 5x5 as four parity phases, output gradient sampled at stride 2 out of its piece planes."""
    from radar_depth_amd import convdesc as cd, ops
    n = 2
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, c, h, w, generator=g)
    wcat = torch.randn(c, c, 5, 5, generator=g, requires_grad=True)
    u = torch.zeros(n, c, 2 * h, 2 * w)
    u[:, :, ::2, ::2] = x
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.upproj_fwd(n, h, w, c, c)
    if not ops.wgrad_split_pre_supported(d):
        pytest.skip("channel count not a multiple of 16")
    slabs = torch.full((ops.wgrad_split_workspace_floats(d),), float("nan"), device="cuda")
    ops.wgrad_split_pre(d, ops.split_pieces(ops.nchw_to_nhwc(x.cuda())), ops.split_pieces(ops.nchw_to_nhwc(gy.cuda())), slabs)
    g_all = torch.full((c, c, 5, 5), float("nan"), device="cuda")
    ops.wgrad_split_reduce(d, slabs, g_all[:c // 2], co_off=0)
    ops.wgrad_split_reduce(d, slabs, g_all[c // 2:], co_off=c // 2)
    torch.cuda.synchronize()
    assert not torch.isnan(g_all).any()
    assert _rel(g_all.cpu(), wcat.grad) < 5e-5


def test_wgrad_split_pre_launches_are_bitwise_reproducible():
    from radar_depth_amd import convdesc as cd, ops
    from radar_depth_amd._lib import lib
    g = torch.Generator().manual_seed(6)
    n, ci, co, h, w = 4, 128, 128, 57, 100
    d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    xp = ops.split_pieces(torch.randn(n, h, w, ci, generator=g).cuda())
    yp = ops.split_pieces(torch.randn(n, h, w, co, generator=g).cuda())
    first = None
    for it in range(60):
        lib().rd_debug_poison_lds(ops.current_stream())
        slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
        ops.wgrad_split_pre(d, xp, yp, slabs)
        grad = torch.empty(co, ci, 3, 3, device="cuda")
        ops.wgrad_split_reduce(d, slabs, grad)
        if first is None:
            first = grad.clone()
        else:
            assert torch.equal(grad, first), it
    torch.cuda.synchronize()
