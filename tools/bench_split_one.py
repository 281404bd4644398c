#!/usr/bin/env python3
"""Timing of rd_gconv_split on a few layers (diagnostics: RD_GCONV_SPLIT_DEBUG / RD_GCONV_SPLIT_FORCE experiments)."""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops
from bench_ops import timeit
from bench_split import plan
B = 16
for name, ci, co, k, h, w in [("layer1", 64, 64, 3, 113, 200), ("layer2", 128, 128, 3, 57, 100), ("layer3", 256, 256, 3, 29, 50), ("layer4", 512, 512, 3, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda"); wt = torch.randn(co, ci, k, k, device="cuda")
    y = torch.empty(B, h, w, co, device="cuda")
    ws = ops.pack_weights_split(wt)
    t = timeit(lambda: ops.gconv_split(d, x, ws, y))
    fl = 2.0 * B * h * w * ci * co * k * k
    print("%-8s %8.1f us %6.1f TF | %s" % (name, t * 1e6, fl / t / 1e12, plan(d)))
