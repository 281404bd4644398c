#!/usr/bin/env python3
"""gconv_sp2_kernel: start offset of the co-resident workgroup (RD_GCONV_SP2_STAGGER, units of 64 clocks).  python tools/sweep_sp2_stagger.py"""
import os
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = 16
for name, ci, co, k, h, w in [("layer1", 64, 64, 3, 113, 200), ("layer2", 128, 128, 3, 57, 100), ("layer3", 256, 256, 3, 29, 50), ("layer4", 512, 512, 3, 15, 25),
                              ("dec2c2", 64, 64, 3, 60, 100)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda")
    y = torch.empty(B, h, w, co, device="cuda")
    ws = ops.pack_weights_split(torch.randn(co, ci, k, k, device="cuda"))
    xp = ops.split_pieces(x)
    out = []
    for st in (0, 16, 32, 64, 96, 128, 192):
        os.environ["RD_GCONV_SP2_STAGGER"] = str(st)
        out.append("%d: %.1f" % (st * 64, timeit(lambda: ops.gconv_split_pre(d, xp, ws, y)) * 1e6))
    os.environ["RD_GCONV_SP2_STAGGER"] = "0"
    print("%-7s us by start offset (clocks): %s" % (name, " | ".join(out)), flush=True)
