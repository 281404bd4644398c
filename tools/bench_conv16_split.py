#!/usr/bin/env python3
"""The 16 -> 16 channel 3x3 layers at the bench geometry: rd_gconv (16x16x4 fp32 MFMA, csrc/conv16.hip) vs rd_conv16_split (three-piece
operands on v_mfma_f32_16x16x32_bf16).   python tools/bench_conv16_split.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from bench_ops import timeit  # noqa: E402

L = lib()
for name, b, h, w in (("depth layer1 113x200", 16, 113, 200), ("dec4 conv2 240x400", 16, 240, 400)):
    for direction in ("fwd", "dgrad"):
        d = cd.conv_fwd(b, h, w, 16, 16, 3, 1, 1) if direction == "fwd" else cd.conv_dgrad(b, h, w, 16, 16, 3, 1, 1)[0]
        x = torch.randn(b, h, w, 16, device="cuda")
        wt = torch.randn(16, 16, 3, 3, device="cuda")
        wp = ops.pack_weights(wt, transpose=direction == "dgrad")
        out = torch.empty(b, h, w, 16, device="cuda")
        tiles = L.rd_gconv_stat_tiles_ws(C.byref(d))
        st = torch.zeros(tiles, 2, 16, device="cuda")
        t32 = timeit(lambda: ops.gconv(d, x, wp, out, stat=st))
        tsp = timeit(lambda: check(L.rd_conv16_split(C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, ptr(st), current_stream()), "x"))
        gf = 2.0 * b * h * w * 16 * 16 * 9 / 1e9
        print("%-22s %-5s %6.2f GF | fp32 %6.1f us %5.1f TF | split %6.1f us %5.1f TF  x%.2f" % (name, direction, gf, t32 * 1e6, gf / t32 / 1e3, tsp * 1e6, gf / tsp / 1e3, t32 / tsp))
