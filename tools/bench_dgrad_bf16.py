#!/usr/bin/env python3
"""rd_gconv_bf16 on the input-gradient descriptors whose shape differs from a forward conv (stride-2 parity phases, 1x1 stride 2,
UpProj) at B=16 450x800: time, algorithmic GB/s (fp32 dy read + dx written), plan.   python tools/bench_dgrad_bf16.py"""
import sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops
sys.argv = sys.argv[:1]
from tools.bench_ops import timeit
from tools.bench_ops_bf16 import plan
B = 16
rows = []
for name, (ci, co, k, s, p, h, w) in {"l2.0 s2 dgrad": (64, 128, 3, 2, 1, 113, 200), "l3.0 s2 dgrad": (128, 256, 3, 2, 1, 57, 100),
                                      "l4.0 s2 dgrad": (256, 512, 3, 2, 1, 29, 50), "ds1x1 s2 dgrad": (64, 128, 1, 2, 0, 113, 200),
                                      "d.l2.0 s2 dgrad": (16, 32, 3, 2, 1, 113, 200)}.items():
    d, zero_fill = cd.conv_dgrad(B, h, w, ci, co, k, s, p)
    rows.append((name, d, torch.randn(co, ci, k, k, device="cuda")))
for c, h, w in [(256, 15, 25), (128, 30, 50), (64, 60, 100), (32, 120, 200)]:
    rows.append(("upproj dgrad %d" % c, cd.upproj_dgrad(B, h, w, c, c), torch.randn(c, c, 5, 5, device="cuda")))
for name, d, wt in rows:
    dy = torch.randn(B, d.Hi, d.Wi, d.Cin, device="cuda")
    dx = torch.zeros(B, d.Ho, d.Wo, d.Cout, device="cuda")
    wd = ops.pack_weights_bf16(wt, transpose=True)
    t = timeit(lambda: ops.gconv_bf16(d, dy, wd, dx))
    byts = 4.0 * (dy.numel() + dx.numel())
    print("%-18s %7.1f MB | %7.1f us %6.0f GB/s | %s" % (name, byts / 1e6, t * 1e6, byts / t / 1e9, plan(d)))
