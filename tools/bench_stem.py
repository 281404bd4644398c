"""Timing of the 7x7/2 stem kernels (forward, weight gradient) at the bench geometry: python tools/bench_stem.py"""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from tools.bench_ops import timeit
L = lib()
N, H, W = 16, 450, 800
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
x = torch.rand(N, 4, H, W, device="cuda")
hw = H * W
for cin, cout, first in ((3, 64, 0), (1, 16, 3)):
    planes = (C.c_void_p * 3)(*[x.data_ptr() + 4 * hw * (first + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[4 * hw if c < cin else 0 for c in range(3)])
    w = torch.randn(cout, cin, 7, 7, device="cuda") * 0.1
    wp = w.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous()
    out = torch.empty(N, Ho, Wo, cout, device="cuda")
    stat = torch.empty(L.rd_stem_stat_tiles(N, H, W), 2, cout, device="cuda")
    t = timeit(lambda: check(L.rd_stem_fwd(planes, strides, cin, N, H, W, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), "stem_fwd"))
    gf = 2.0 * N * Ho * Wo * cout * 49 * cin / 1e9
    print("stem fwd  %d->%d: %7.1f us  %5.1f TF (%4.1f%% of fp32 peak)" % (cin, cout, t * 1e6, gf / t / 1e3, 100 * gf / t / 1e3 / 157.3))
    L.rd_stem_wgrad_workspace_floats.restype = C.c_int64
    ws = torch.empty(int(L.rd_stem_wgrad_workspace_floats(N, H, W, cin, cout)), device="cuda")
    gw = torch.empty(cout, cin, 7, 7, device="cuda")
    dout = torch.randn(N, Ho, Wo, cout, device="cuda")
    t = timeit(lambda: check(L.rd_stem_wgrad(planes, strides, cin, N, H, W, ptr(dout), cout, ptr(gw), ptr(ws), current_stream()), "stem_wgrad"))
    print("stem wgrad %d->%d: %7.1f us  %5.1f TF (%4.1f%% of fp32 peak)" % (cin, cout, t * 1e6, gf / t / 1e3, 100 * gf / t / 1e3 / 157.3))
