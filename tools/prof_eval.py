#!/usr/bin/env python3
"""One eval-mode configuration for rocprofv3: python tools/prof_eval.py <fp32|bf16> <batch> [calls]  (eager launches)."""
import sys, torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipInference
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch
ops, b = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
m = ResNet_latefusion(18, "upproj", [450, 800], 4, False).cuda()
inf = HipInference(m, b, 450, 800, use_graph=False, operands=ops)
x, _ = make_batch(b, 450, 800, 1)
x = x.cuda()
for _ in range(n):
    inf(x)
torch.cuda.synchronize()
