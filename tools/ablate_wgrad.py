import sys, torch
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
B, dev = 16, "cuda"
for (ci, co, k, s, p, h, w) in [(64, 64, 3, 1, 1, 113, 200), (256, 256, 3, 1, 1, 29, 50), (512, 512, 3, 1, 1, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
    x = torch.randn(B, h, w, ci, device=dev); y = torch.randn(B, d.Ho, d.Wo, co, device=dev)
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device=dev)
    for _ in range(3): ops.wgrad(d, x, y, slabs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.wgrad(d, x, y, slabs)
    e1.record(); torch.cuda.synchronize()
    fl = 2.0 * B * d.Ho * d.Wo * co * ci * k * k
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print("%s  %.1f us  %.1f TF" % ((ci, co, h, w), t * 1e6, fl / t / 1e12))
