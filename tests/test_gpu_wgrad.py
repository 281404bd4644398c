"""GPU parity of the MFMA weight-gradient kernel (rd_wgrad + rd_wgrad_reduce) against torch CPU autograd.
Tolerance 5e-5 of the gradient's max magnitude (split-K over up to ~360k pixels, fp32 accumulate)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 3, 1, 1, 113, 200),
    (2, 64, 128, 3, 2, 1, 113, 200),
    (2, 128, 128, 3, 1, 1, 57, 100),
    (2, 64, 128, 1, 2, 0, 113, 200),
    (2, 512, 512, 3, 1, 1, 15, 25),
    (2, 640, 512, 1, 1, 0, 15, 25),
    (2, 16, 16, 3, 1, 1, 113, 200),
    (2, 16, 32, 3, 2, 1, 113, 200),
    (2, 16, 32, 1, 2, 0, 113, 200),
    (2, 32, 32, 3, 1, 1, 57, 100),
    (1, 16, 16, 3, 1, 1, 240, 400),
    (3, 32, 48, 3, 1, 1, 9, 7),
    # column-strip kernel edge cases: ragged last strip (26 = 25 + 1 columns), odd row counts, one-row segments, partial
    # 64-channel blocks, a single image
    (2, 64, 64, 3, 1, 1, 25, 26),
    (1, 96, 160, 3, 1, 1, 31, 51),
    (5, 128, 64, 3, 1, 1, 24, 3),
    (2, 64, 64, 3, 1, 1, 29, 75),
    # immediate-offset tiled kernel (short image -> no strip) with ragged tiles
    (2, 64, 96, 3, 1, 1, 7, 33),
    (2, 256, 64, 3, 2, 1, 21, 27),
    # csrc/wgrad16.hip (16 -> 16 channels on the 16x16x4 MFMA): partial 16x16 tiles, images smaller than a tile, many tiles per split
    (3, 16, 16, 3, 1, 1, 37, 45),
    (5, 16, 16, 3, 1, 1, 1, 1),
    (2, 16, 16, 3, 1, 1, 3, 50),
    (16, 16, 16, 3, 1, 1, 113, 200),
    # csrc/wgrad1x1.hip (one tap, >= 64 channels each side): every tile shape, partial channel blocks, pixel counts that are not a
    # multiple of the 32-pixel chunk, stride-2 gathers over odd image sizes, fewer pixels than one chunk
    (2, 128, 256, 1, 2, 0, 57, 100),
    (4, 256, 512, 1, 2, 0, 29, 50),
    (3, 512, 256, 1, 1, 0, 15, 25),
    (1, 64, 64, 1, 1, 0, 3, 5),
    (2, 96, 160, 1, 2, 0, 9, 7),
    (5, 192, 64, 1, 1, 0, 7, 9),
    (16, 640, 512, 1, 1, 0, 15, 25),
])
def test_wgrad_conv(cfg):
    from radar_depth_amd import convdesc as cd, ops
    n, ci, co, k, s, p, h, w = cfg
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g, requires_grad=True)
    y = F.conv2d(x, wt, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    ops.wgrad(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs)
    grad = torch.full((co, ci, k, k), float("nan"), device="cuda")
    ops.wgrad_reduce(d, slabs, grad)
    torch.cuda.synchronize()
    assert _rel(grad.cpu(), wt.grad) < 5e-5, cfg
    ops.wgrad_reduce(d, slabs, grad, accumulate=True)
    torch.cuda.synchronize()
    assert _rel(grad.cpu(), 2 * wt.grad) < 5e-5


@pytest.mark.parametrize("c,h,w", [(256, 15, 25), (64, 60, 100), (32, 13, 9), (64, 27, 26), (128, 25, 51), (64, 5, 3)])
def test_wgrad_upproj(c, h, w):
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
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    ops.wgrad(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs)
    g_up = torch.empty(c // 2, c, 5, 5, device="cuda")
    g_bt = torch.empty(c // 2, c, 5, 5, device="cuda")
    ops.wgrad_reduce(d, slabs, g_up, co_off=0)
    ops.wgrad_reduce(d, slabs, g_bt, co_off=c // 2)
    torch.cuda.synchronize()
    assert _rel(g_up.cpu(), wcat.grad[:c // 2]) < 5e-5
    assert _rel(g_bt.cpu(), wcat.grad[c // 2:]) < 5e-5


def test_wgrad_reduce_batched_is_bit_identical():
    """rd_wgrad_reduce_batched (all slab reductions of a backward segment in two launches) against one rd_wgrad_reduce per weight
    tensor: same summation order, hence the same bits -- a many-split layer (two-stage reduction), a few-split layer (stage 2
    reads the slabs), the 16-channel kernel's slabs, and an UpProj pair (two column ranges of one slab set)."""
    from radar_depth_amd import convdesc as cd, ops
    g = torch.Generator().manual_seed(5)
    jobs, want = [], []
    for n, ci, co, k, s, p, h, w in [(4, 64, 64, 3, 1, 1, 113, 200), (2, 512, 512, 3, 1, 1, 15, 25), (4, 16, 16, 3, 1, 1, 113, 200),
                                     (2, 64, 128, 1, 2, 0, 113, 200)]:
        d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
        x = torch.randn(n, h, w, ci, generator=g).cuda()
        gy = torch.randn(n, d.Ho, d.Wo, co, generator=g).cuda()
        slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
        ops.wgrad(d, x, gy, slabs)
        ref = torch.empty(co, ci, k, k, device="cuda")
        ops.wgrad_reduce(d, slabs, ref)
        jobs.append((d, slabs, torch.full((co, ci, k, k), float("nan"), device="cuda"), 0))
        want.append(ref)
    c, h, w = 64, 60, 100
    d = cd.upproj_fwd(2, h, w, c, c)
    x = torch.randn(2, h, w, c, generator=g).cuda()
    gy = torch.randn(2, 2 * h, 2 * w, c, generator=g).cuda()
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    ops.wgrad(d, x, gy, slabs)
    for off in (0, c // 2):
        ref = torch.empty(c // 2, c, 5, 5, device="cuda")
        ops.wgrad_reduce(d, slabs, ref, co_off=off)
        jobs.append((d, slabs, torch.full((c // 2, c, 5, 5), float("nan"), device="cuda"), off))
        want.append(ref)
    nb1, nb2 = ops.wgrad_reduce_batched(jobs)
    torch.cuda.synchronize()
    assert nb1 > 0 and nb2 > 0
    for (_, _, got, _), ref in zip(jobs, want):
        assert torch.equal(got, ref)
