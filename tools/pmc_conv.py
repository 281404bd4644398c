"""Tiny driver for rocprofv3 --pmc runs: a few launches of the dominant conv shapes (layer1 3x3 64->64, B=16)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radar_depth_amd import convdesc as cd, ops
B = 16
dev = "cuda"
# (round 2: + a stride-2 3x3 forward, which runs as input-parity groups)
for (ci, co, k, s, p, h, w) in [(64, 64, 3, 1, 1, 113, 200), (128, 128, 3, 1, 1, 57, 100), (512, 512, 3, 1, 1, 15, 25), (64, 128, 3, 2, 1, 113, 200)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
    x = torch.randn(B, h, w, ci, device=dev)
    wt = torch.randn(co, ci, k, k, device=dev)
    wp = ops.pack_weights(wt)
    y = torch.empty(B, d.Ho, d.Wo, co, device=dev)
    stat = torch.zeros(ops.gconv_stat_tiles(d), 2, co, device=dev)
    slabs = torch.empty(ops.wgrad_workspace_floats(d), device=dev)
    for _ in range(3):
        ops.gconv(d, x, wp, y, stat=stat)
        if s == 1:
            ops.wgrad(d, x, y, slabs)
torch.cuda.synchronize()
