import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
import ctypes as C
from radar_depth_amd._lib import lib
n, ci, co, k, s, h, w = 2, 64, 64, 3, 1, 6, 7
g = torch.Generator().manual_seed(0)
x = torch.randn(n, ci, h, w, generator=g)
wt = torch.randn(co, ci, k, k, generator=g) * 0.05
y = F.conv2d(x, wt, stride=s, padding=k // 2)
d = cd.conv_fwd(n, h, w, ci, co, k, s, k // 2)
info = (C.c_int32 * 10)()
lib().rd_gconv_plan_info(C.byref(d), info)
print("plan info", list(info), "ws floats", lib().rd_gconv_workspace_floats(C.byref(d)), "stat tiles", ops.gconv_stat_tiles(d))
xs = ops.nchw_to_nhwc(x.cuda())
wp_ = ops.pack_weights(wt.cuda())
for with_stat in (False, True):
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
    stat = torch.full((ops.gconv_stat_tiles(d), 2, co), float("nan"), device="cuda") if with_stat else None
    ops.gconv(d, xs, wp_, out, stat=stat)
    torch.cuda.synchronize()
    err = (ops.nhwc_to_nchw(out).cpu() - y).abs()
    print("stat" if with_stat else "no stat", "max err", err.max().item(), "bad channels", sorted(set((err > 1e-3).nonzero()[:, 1].tolist())))
