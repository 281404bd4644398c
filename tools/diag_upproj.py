import sys, ctypes as C, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import lib
for c, h, w in [(128, 8, 12), (128, 30, 50), (256, 4, 6), (256, 15, 25), (64, 16, 24), (32, 32, 48)]:
    n = 2
    g = torch.Generator().manual_seed(2)
    wcat = torch.randn(c, c, 5, 5, generator=g) * (2.0 / (25 * c)) ** 0.5
    u = torch.zeros(n, c, 2 * h, 2 * w, requires_grad=True)
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    want = u.grad[:, :, ::2, ::2]
    dd = cd.upproj_dgrad(n, h, w, c, c)
    info = (C.c_int32 * 10)()
    lib().rd_gconv_plan_info(C.byref(dd), info)
    wd = ops.pack_weights(wcat.cuda(), transpose=True)
    dx = torch.full((n, h, w, c), float("nan"), device="cuda")
    ops.gconv(dd, ops.nchw_to_nhwc(gy.cuda()), wd, dx)
    torch.cuda.synchronize()
    got = dx.permute(0, 3, 1, 2).cpu()
    err = (got - want).abs()
    print(c, h, w, "plan", list(info), "relerr %.2e" % (err.max() / want.abs().max()).item(),
          "bad px frac %.3f" % (err.amax(1) > 1e-4 * want.abs().max()).float().mean().item())
    if err.max() / want.abs().max() > 1e-4:
        bad = (err.amax(1) > 1e-4 * want.abs().max())[0]
        print(" bad map sample0:\n", bad.int())
        badc = (err.amax((0, 2, 3)) > 1e-4 * want.abs().max())
        print(" bad channels:", badc.nonzero().flatten().tolist()[:40], "count", int(badc.sum()))
