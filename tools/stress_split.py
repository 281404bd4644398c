#!/usr/bin/env python3
"""Race hunt for the split kernels: every launch of a deterministic kernel must reproduce the first launch bit for bit.  Convolution forward
(with BatchNorm partial sums), input gradient, weight gradient (+ slab reduction) on a few shapes, N repetitions each, under a concurrent
HBM-bound kernel on a second stream (different wave arrival orders).  Run on the GPU box:  python tools/stress_split.py [N]"""
import sys

import torch

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
side = torch.cuda.Stream()
junk = torch.randn(64 << 20, device="cuda")
bad = 0
for n, ci, co, k, h, w in [(16, 64, 64, 3, 113, 200), (16, 128, 128, 3, 57, 100), (16, 512, 512, 3, 15, 25), (4, 256, 256, 3, 29, 50), (3, 96, 160, 3, 31, 51)]:
    d = cd.conv_fwd(n, h, w, ci, co, k, 1, 1)
    dd, _ = cd.conv_dgrad(n, h, w, ci, co, k, 1, 1)
    x = torch.randn(n, h, w, ci, device="cuda")
    dy = torch.randn(n, h, w, co, device="cuda")
    wt = torch.randn(co, ci, k, k, device="cuda") * 0.05
    wf, wb = ops.pack_weights_split(wt), ops.pack_weights_split(wt, transpose=True)
    res = {}
    for it in range(N):
        with torch.cuda.stream(side):
            junk.mul_(1.0001)
        out = {}
        if ops.gconv_split_supported(d):
            y = torch.empty(n, h, w, co, device="cuda")
            st = torch.zeros(ops.gconv_split_stat_tiles(d), 2, co, device="cuda")
            ops.gconv_split(d, x, wf, y, stat=st)
            out["fwd"], out["stat"] = y, st
        if ops.gconv_split_supported(dd):
            dx = torch.empty(n, h, w, ci, device="cuda")
            ops.gconv_split(dd, dy, wb, dx)
            out["dgrad"] = dx
        if ops.wgrad_split_supported(d):
            sl = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
            g = torch.empty(co, ci, k, k, device="cuda")
            ops.wgrad_split(d, x, dy, sl)
            ops.wgrad_split_reduce(d, sl, g)
            out["wgrad"] = g
        torch.cuda.synchronize()
        for key, v in out.items():
            if key not in res:
                res[key] = v.clone()
            elif not torch.equal(res[key], v):
                bad += 1
                print("MISMATCH %s %s iteration %d: %d elements differ" % ((n, ci, co, k, h, w), key, it, (res[key] != v).sum().item()))
    print("shape", (n, ci, co, k, h, w), "kernels", sorted(res), "repetitions", N, "mismatches so far", bad)
c, h, w = 128, 30, 50
d = cd.upproj_fwd(4, h, w, c, c)
x = torch.randn(4, h, w, c, device="cuda"); dy = torch.randn(4, 2 * h, 2 * w, c, device="cuda")
ref = None
for it in range(N):
    sl = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    g0 = torch.empty(c // 2, c, 5, 5, device="cuda")
    g = torch.empty(c // 2, c, 5, 5, device="cuda")
    ops.wgrad_split(d, x, dy, sl)
    ops.wgrad_split_reduce(d, sl, g0, co_off=0)          # (column ranges in increasing order: stage 1 of the reduction runs with the first)
    ops.wgrad_split_reduce(d, sl, g, co_off=c // 2)
    torch.cuda.synchronize()
    if ref is None:
        ref = g.clone()
    elif not torch.equal(ref, g):
        bad += 1
        print("MISMATCH upproj wgrad iteration", it)
print("upproj wgrad repetitions", N, "| total mismatches", bad)
