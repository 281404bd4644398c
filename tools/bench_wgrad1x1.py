#!/usr/bin/env python3
"""Weight gradient of the 1x1 layers at the bench geometry (B=16, 450x800): rd_wgrad time and fraction of the fp32 MFMA peak.
Run twice on the GPU box to compare kernels:  python tools/bench_wgrad1x1.py ; RD_WGRAD_NO1X1=1 python tools/bench_wgrad1x1.py"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402

PEAK = 157.3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
LAYERS = [("layer2 ds", 64, 128, 2, 113, 200), ("layer3 ds", 128, 256, 2, 57, 100), ("layer4 ds", 256, 512, 2, 29, 50),
          ("d.layer3 ds", 32, 64, 2, 57, 100), ("d.layer4 ds", 64, 128, 2, 29, 50),
          ("fusion", 640, 512, 1, 15, 25), ("conv2", 512, 256, 1, 15, 25)]
out = {"kernel": "wgrad_kernel<1,...>" if os.environ.get("RD_WGRAD_NO1X1") else "wgrad1x1_kernel", "batch": B, "layers": {}}
tot = 0.0
for name, ci, co, s, h, w in LAYERS:
    d = cd.conv_fwd(B, h, w, ci, co, 1, s, 0)
    x = torch.randn(B, h, w, ci, device="cuda")
    gy = torch.randn(B, d.Ho, d.Wo, co, device="cuda")
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    grad = torch.empty(co, ci, 1, 1, device="cuda")
    for _ in range(5):
        ops.wgrad(d, x, gy, slabs)
        ops.wgrad_reduce(d, slabs, grad)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    n = 50
    e[0].record()
    for _ in range(n):
        ops.wgrad(d, x, gy, slabs)
    e[1].record()
    for _ in range(n):
        ops.wgrad_reduce(d, slabs, grad)
    e[2].record()
    torch.cuda.synchronize()
    tw, tr = e[0].elapsed_time(e[1]) / n * 1e3, e[1].elapsed_time(e[2]) / n * 1e3
    fl = 2.0 * B * d.Ho * d.Wo * ci * co
    out["layers"][name] = {"wgrad_us": round(tw, 1), "reduce_us": round(tr, 1), "tflops": round(fl / tw / 1e6, 1),
                           "frac_fp32_peak": round(fl / tw / 1e6 / PEAK, 3)}
    tot += tw + tr
    print("%-12s %4d->%4d s%d  wgrad %7.1f us  reduce %6.1f us  %6.1f TF (%.3f of peak)" % (name, ci, co, s, tw, tr, fl / tw / 1e6, fl / tw / 1e6 / PEAK))
out["total_us"] = round(tot, 1)
print(json.dumps(out))
