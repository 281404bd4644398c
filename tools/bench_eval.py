#!/usr/bin/env python3
"""Eval-mode (validate(), main.py:564-595) latency of the HIP inference path: batch 1 and 16, 450x800, hipGraph replay."""
import sys, time, torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipInference
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch
torch.manual_seed(0)
m = ResNet_latefusion(18, "upproj", [450, 800], 4, False).cuda()
ops = sys.argv[1] if len(sys.argv) > 1 else "fp32"     # fp32 | bf16 (conv operands; csrc/gconv_bf16.hip)
print("conv operands:", ops)
for b, g in ((1, True), (1, False), (16, True), (16, False)):
    inf = HipInference(m, b, 450, 800, use_graph=g, operands=ops)
    x, _ = make_batch(b, 450, 800, 1)
    x = x.cuda()
    for _ in range(5): inf(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n): inf(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(("hipGraph " if g else "eager    ") + "eval forward b=%d: %.3f ms/call  %.1f samples/s  (35.47 GFLOP/sample algorithmic -> %.1f TFLOP/s)" % (b, dt * 1e3, b / dt, 35.47e9 * b / dt / 1e12))
