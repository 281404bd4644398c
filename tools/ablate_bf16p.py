#!/usr/bin/env python3
"""Ablations of the persistent bf16-storage convolution (RD_GCONV_BF16P_DEBUG bits; results garbage): which part of a launch the time is.
    python tools/ablate_bf16p.py [layer ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from bench_ops import timeit  # noqa: E402

L = lib()
B = 16
LAYERS = {"layer1": (64, 64, 113, 200), "layer2": (128, 128, 57, 100), "layer3": (256, 256, 29, 50), "layer4": (512, 512, 15, 25), "dec3c2": (32, 32, 120, 200)}
VARIANTS = [("full", 0), ("noEPI", 16), ("noMFMA", 1), ("noPcopy", 8), ("noWcopy", 4), ("noW+noP", 12), ("noW+noP+noEPI", 28), ("noMFMA+noEPI", 17), ("none", 29)]
for name in (sys.argv[1:] or list(LAYERS)):
    ci, co, h, w = LAYERS[name]
    d = cd.conv_fwd(B, h, w, ci, co, 3, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda").to(torch.bfloat16)
    wp = ops.pack_weights_bf16(torch.randn(co, ci, 3, 3, device="cuda"))
    y = torch.empty(B, h, w, co, device="cuda", dtype=torch.bfloat16)
    row = []
    for tag, bits in VARIANTS:
        os.environ["RD_GCONV_BF16P_DEBUG"] = str(bits)
        t = timeit(lambda: check(L.rd_gconv_bf16_t(1, C.byref(d), ptr(x), ptr(wp), ptr(y), None, 0, 0, None, 0, None, current_stream()), "gconv"))
        row.append("%s %.1f" % (tag, t * 1e6))
    os.environ["RD_GCONV_BF16P_DEBUG"] = "0"
    for stg in (0, 8, 16, 32, 64, 128):
        os.environ["RD_GCONV_BF16P_STAGGER"] = str(stg)
        t = timeit(lambda: check(L.rd_gconv_bf16_t(1, C.byref(d), ptr(x), ptr(wp), ptr(y), None, 0, 0, None, 0, None, current_stream()), "gconv"))
        row.append("stagger%d %.1f" % (stg, t * 1e6))
    del os.environ["RD_GCONV_BF16P_STAGGER"]
    print("%-8s us: %s" % (name, " | ".join(row)), flush=True)
