import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
for c, h, w in [(256, 15, 25), (64, 60, 100), (32, 13, 9), (64, 6, 13), (64, 3, 13), (64, 3, 12)]:
    n = 2
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, c, h, w, generator=g)
    wcat = torch.randn(c, c, 5, 5, generator=g, requires_grad=True)
    u = torch.zeros(n, c, 2 * h, 2 * w); u[:, :, ::2, ::2] = x
    y = F.conv2d(u, wcat, padding=2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    d = cd.upproj_fwd(n, h, w, c, c)
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    ops.wgrad(d, ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda()), slabs)
    gr = torch.empty(c, c, 5, 5, device="cuda")
    ops.wgrad_reduce(d, slabs, gr)
    torch.cuda.synchronize()
    err = (gr.cpu() - wcat.grad).abs().amax((0, 1)) / wcat.grad.abs().max()
    print(c, h, w, "per-tap max rel err:\n", (err * 1e3).round().int())
    e2 = (gr.cpu() - wcat.grad).abs().amax((2, 3))
    print(" bad (o,i) fraction", (e2 > 1e-3).float().mean().item(), "bad o blocks", (e2 > 1e-3).float().mean(1).reshape(-1, 32).mean(1)[:8], "bad i blocks", (e2 > 1e-3).float().mean(0).reshape(-1, 32).mean(1)[:8])
