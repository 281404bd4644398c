#!/usr/bin/env python3
"""Ablations of wgrad_split_kernel's staging waves (RD_WGRAD_SPLIT_DEBUG bits: 1 no split arithmetic, 2 no LDS stores, 4 no global loads):
is the weight gradient bound by its staging waves, and by which part of them?  Results are garbage; only the times mean something."""
import os
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

B = 16
for name, ci, co, h, w in [("layer1", 64, 64, 113, 200), ("layer2", 128, 128, 57, 100), ("layer3", 256, 256, 29, 50), ("layer4", 512, 512, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, 3, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda")
    dy = torch.randn(B, h, w, co, device="cuda")
    slabs = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    out = []
    for m, tag in ((0, "full"), (1, "no split arithmetic"), (2, "no LDS stores"), (3, "neither"), (4, "no global loads"), (7, "no staging at all")):
        os.environ["RD_WGRAD_SPLIT_DEBUG"] = str(m)
        out.append("%s %.1f" % (tag, timeit(lambda: ops.wgrad_split(d, x, dy, slabs)) * 1e6))
    os.environ["RD_WGRAD_SPLIT_DEBUG"] = "0"
    print("%-7s us: %s" % (name, " | ".join(out)), flush=True)
