#!/usr/bin/env python3
"""Ablations of the Winograd split kernel (RD_WINO_DEBUG bits: 1 no MFMAs, 2 no split + A stores, 4 no weight copies, 8 no epilogue; results
are garbage then): one process per setting, layer1 / layer3 / layer4 shapes at b = 16.   python tools/ablate_wino.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, ".")
    sys.path.insert(0, "tools")
    from radar_depth_amd import ops
    from bench_ops import timeit
    out = []
    for (c, h, w) in ((64, 113, 200), (256, 29, 50), (512, 15, 25)):
        x = torch.randn(16, h, w, c, device="cuda")
        wt = torch.randn(c, c, 3, 3, device="cuda") * 0.05
        y = torch.empty(16, h, w, c, device="cuda")
        u = ops.wino_pack(wt)
        out.append("%7.1f" % (1e6 * timeit(lambda: ops.wino_conv3x3(x, u, y))))
    print(" ".join(out))
    sys.exit(0)
print("us per launch at b=16:            64ch 113x200 | 256ch 29x50 | 512ch 15x25")
for dbg, what in ((0, "full kernel"), (1, "no MFMAs"), (2, "no split / A stores"), (4, "no weight copies"), (8, "no epilogue"), (3, "no MFMAs, no A stores"),
                  (6, "no A stores, no weight copies"), (7, "loads + transform + barriers only"), (15, "skeleton (loads + adds + barriers)"), (9, "no MFMAs, no epilogue"),
                  (12, "no weight copies, no epilogue"), (16, "chunk rotation on"), (32, "staging waves idle (barriers only)"), (40, "staging idle, no epilogue"), (64, "compute waves idle (barriers only)"), (72, "compute idle, no epilogue"), (104, "barriers only"), (128, "pixel-block-major order at >= 256 channels"), (66, "compute idle, no split / A stores"), (320, "compute idle, no patch loads"),
                  (322, "compute idle, no patch loads, no split / stores"), (256, "no patch loads")):
    r = subprocess.run([sys.executable, __file__, "--one"], env=dict(os.environ, RD_WINO_DEBUG=str(dbg)), capture_output=True, text=True)
    print("dbg %2d %-38s %s" % (dbg, what, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]), flush=True)
