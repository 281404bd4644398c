#!/usr/bin/env python3
"""rd_wgrad (fp32 MFMA) vs rd_wgrad_split per layer at the bench geometry (B=16, 450x800): the 3x3 / stride-1 layers with >= 64 channels.
Run on the GPU box:  python tools/bench_wgrad_split.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import lib  # noqa: E402
from bench_ops import CONVS, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
tot = [0.0, 0.0]
for name, cnt, ci, co, k, s, p, h, w in CONVS:
    d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
    if not ops.wgrad_split_supported(d):
        continue
    x = torch.randn(B, h, w, ci, device="cuda")
    y = torch.randn(B, d.Ho, d.Wo, co, device="cuda")
    s1 = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    s2 = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    g = torch.empty(co, ci, k, k, device="cuda")
    t1 = timeit(lambda: ops.wgrad(d, x, y, s1))
    r1 = timeit(lambda: ops.wgrad_reduce(d, s1, g))
    t2 = timeit(lambda: ops.wgrad_split(d, x, y, s2))
    r2 = timeit(lambda: ops.wgrad_split_reduce(d, s2, g))
    if ops.wgrad_split_pre_supported(d):
        xp, yp = ops.split_pieces(x), ops.split_pieces(y)
        t3 = timeit(lambda: ops.wgrad_split_pre(d, xp, yp, s2))
        print("%-18s pre-split operands: %7.1f us %6.1f TF  x%.2f vs split" % (name, t3 * 1e6, 2.0 * B * d.Ho * d.Wo * ci * co * k * k / t3 / 1e12, t2 / t3))
        tot.append(cnt * (t3 + r2)) if len(tot) == 2 else tot.__setitem__(2, tot[2] + cnt * (t3 + r2))
        del xp, yp
    v = (C.c_int32 * 4)()
    lib().rd_wgrad_split_plan_info(C.byref(d), v)
    fl = 2.0 * B * d.Ho * d.Wo * ci * co * k * k
    print("%-18s x%d %6.2f GF | fp32 %7.1f us %6.1f TF (+reduce %5.1f) | split %7.1f us %6.1f TF (+reduce %5.1f) x%.2f | splits %d x %d tiles, %d wg"
          % (name, cnt, fl / 1e9, t1 * 1e6, fl / t1 / 1e12, r1 * 1e6, t2 * 1e6, fl / t2 / 1e12, r2 * 1e6, t1 / t2, v[0], v[1], v[2]))
    tot[0] += cnt * (t1 + r1)
    tot[1] += cnt * (t2 + r2)
from bench_ops import UPPROJ  # noqa: E402
for name, c, h, w in UPPROJ:
    d = cd.upproj_fwd(B, h, w, c, c)
    if not ops.wgrad_split_supported(d):
        continue
    x = torch.randn(B, h, w, c, device="cuda")
    y = torch.randn(B, 2 * h, 2 * w, c, device="cuda")
    s1 = torch.empty(ops.wgrad_workspace_floats(d), device="cuda")
    s2 = torch.empty(ops.wgrad_split_workspace_floats(d), device="cuda")
    g = torch.empty(c // 2, c, 5, 5, device="cuda")

    def red1():
        ops.wgrad_reduce(d, s1, g, co_off=0)
        ops.wgrad_reduce(d, s1, g, co_off=c // 2)

    def red2():
        ops.wgrad_split_reduce(d, s2, g, co_off=0)
        ops.wgrad_split_reduce(d, s2, g, co_off=c // 2)
    t1 = timeit(lambda: ops.wgrad(d, x, y, s1))
    r1 = timeit(red1)
    t2 = timeit(lambda: ops.wgrad_split(d, x, y, s2))
    r2 = timeit(red2)
    fl = 2.0 * B * h * w * c * c * 25
    if ops.wgrad_split_pre_supported(d):
        xp, yp = ops.split_pieces(x), ops.split_pieces(y)
        t3 = timeit(lambda: ops.wgrad_split_pre(d, xp, yp, s2))
        print("%-18s pre-split operands: %7.1f us %6.1f TF  x%.2f vs split" % (name, t3 * 1e6, fl / t3 / 1e12, t2 / t3))
        tot.append(t3 + r2) if len(tot) == 2 else tot.__setitem__(2, tot[2] + t3 + r2)
        del xp, yp
    print("%-18s x1 %6.2f GF | fp32 %7.1f us %6.1f TF (+reduce %5.1f) | split %7.1f us %6.1f TF (+reduce %5.1f) x%.2f"
          % (name, fl / 1e9, t1 * 1e6, fl / t1 / 1e12, r1 * 1e6, t2 * 1e6, fl / t2 / 1e12, r2 * 1e6, t1 / t2))
    tot[0] += t1 + r1
    tot[1] += t2 + r2
print("TOTAL weight gradients of these layers (kernel + slab reduction): fp32 MFMA %.2f ms, split %.2f ms%s"
      % (tot[0] * 1e3, tot[1] * 1e3, ", pre-split operands %.2f ms" % (tot[2] * 1e3) if len(tot) > 2 else ""))
