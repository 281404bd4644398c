"""Timing of the head kernels (conv3 16 -> 1 forward / backward, bilinear) at the bench geometry: python tools/bench_head.py"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from tools.bench_ops import timeit
L = lib()
L.rd_head_conv_bwd_workspace_floats.restype = C.c_int64
N, H, W, Cc, Ho, Wo = 16, 240, 400, 16, 450, 800
x = torch.randn(N, H, W, Cc, device="cuda")
w = torch.randn(1, Cc, 3, 3, device="cuda")
d = torch.empty(N, H, W, device="cuda")
dd = torch.randn(N, H, W, device="cuda")
dx = torch.empty(N, H, W, Cc, device="cuda")
dw = torch.empty(1, Cc, 3, 3, device="cuda")
ws = torch.empty(int(L.rd_head_conv_bwd_workspace_floats(N, H, W, Cc)), device="cuda")
pred = torch.empty(N, 1, Ho, Wo, device="cuda")
mb = x.numel() * 4 / 1e6
t = timeit(lambda: check(L.rd_head_conv_fwd(ptr(x), Cc, ptr(w), N, H, W, Cc, ptr(d), current_stream()), "fwd"))
print("head conv fwd : %6.1f us  (%.0f MB in -> %.2f TB/s)" % (t * 1e6, mb, mb / t / 1e6))
t = timeit(lambda: check(L.rd_head_conv_bwd(ptr(x), Cc, ptr(w), ptr(dd), N, H, W, Cc, ptr(dx), Cc, ptr(dw), ptr(ws), current_stream()), "bwd"))
print("head conv bwd : %6.1f us  (dgrad writes %.0f MB, wgrad reads %.0f MB -> %.2f TB/s)" % (t * 1e6, mb, mb, 2 * mb / t / 1e6))
t = timeit(lambda: check(L.rd_bilinear_fwd(ptr(d), N, H, W, ptr(pred), Ho, Wo, current_stream()), "bil"))
print("bilinear fwd  : %6.1f us" % (t * 1e6))
t = timeit(lambda: check(L.rd_bilinear_bwd(ptr(pred), N, Ho, Wo, ptr(dd), H, W, current_stream()), "bilb"))
print("bilinear bwd  : %6.1f us" % (t * 1e6))
