#!/usr/bin/env python3
"""RGB stem forward at the bench geometry (b=16, 3 x 450 x 800 -> 64 x 225 x 400): rd_stem_fwd (fp32 MFMA) vs rd_stem_fwd_bf16 vs
rd_stem_fwd_split (three-piece operands, 8-wave role split).  RD_STEM_SPLIT_DEBUG = 1 / 2 / 4 ablates the split kernel's MFMA walk / patch
staging / output stores (results garbage, times only).   python tools/bench_stem_split.py"""
import ctypes as C, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from bench_ops import timeit
L = lib()
n, cin, cout, h, w = 16, 3, 64, 450, 800
x = torch.randn(n, cin, h, w, device="cuda")
wp = torch.randn(49, cin, cout, device="cuda") * 0.1
hw = h * w
planes = (C.c_void_p * 3)(*[x.data_ptr() + 4 * hw * c for c in range(3)])
strides = (C.c_int64 * 3)(*[cin * hw] * 3)
ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
out = torch.empty(n, ho, wo, cout, device="cuda")
stat = torch.zeros(L.rd_stem_stat_tiles(n, h, w), 2, cout, device="cuda")
for name, fn in (("fp32", L.rd_stem_fwd), ("bf16", L.rd_stem_fwd_bf16), ("split", L.rd_stem_fwd_split)):
    t = timeit(lambda: check(fn(planes, strides, cin, n, h, w, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), name))
    print("%-6s %7.1f us" % (name, t * 1e6))
