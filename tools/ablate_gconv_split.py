#!/usr/bin/env python3
"""Ablations of gconv_split_kernel (RD_GCONV_SPLIT_DEBUG bits: 1 no MFMAs, 2 no fragment reads, 4 no weight copies, 8 no patch
copies / staging, 16 no epilogue): what each part of the kernel costs when the others are (not) there.  Results are garbage by
construction; only the times mean something.   python tools/ablate_gconv_split.py [--pre]"""
import os
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

PRE = "--pre" in sys.argv
B = 16
MASKS = [0, 16, 1, 2, 3, 4, 8, 12, 1 | 16, 2 | 16, 3 | 16, 4 | 8 | 16, 1 | 2 | 4 | 8, 31] if "--all" in sys.argv else [0, 16, 1, 2, 3, 12, 3 | 16, 12 | 16, 31]
NAMES = {1: "noMFMA", 2: "noREAD", 4: "noWcopy", 8: "noPcopy", 16: "noEPI"}


def label(m):
    return "+".join(v for k, v in NAMES.items() if m & k) or "full"


for name, ci, co, k, h, w in [("layer1", 64, 64, 3, 113, 200), ("layer2", 128, 128, 3, 57, 100), ("layer3", 256, 256, 3, 29, 50), ("layer4", 512, 512, 3, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda")
    wt = torch.randn(co, ci, k, k, device="cuda")
    y = torch.empty(B, h, w, co, device="cuda")
    ws = ops.pack_weights_split(wt)
    xp = ops.split_pieces(x) if PRE else None
    fn = (lambda: ops.gconv_split_pre(d, xp, ws, y)) if PRE else (lambda: ops.gconv_split(d, x, ws, y))
    out = []
    for m in MASKS:
        os.environ["RD_GCONV_SPLIT_DEBUG"] = str(m)
        out.append("%s %.1f" % (label(m), timeit(fn) * 1e6))
    os.environ["RD_GCONV_SPLIT_DEBUG"] = "0"
    print("%-7s %s us: %s" % (name, "PRE" if PRE else "split", " | ".join(out)), flush=True)
