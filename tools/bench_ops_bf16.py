#!/usr/bin/env python3
"""Per-layer microbenchmark of the bf16-operand convolution (rd_gconv_bf16) at the bench geometry (B=16, 450x800): time, plan,
TFLOP/s and the algorithmic HBM rate (fp32 input read once + fp32 output written once) -- the bf16 layers are HBM-bound
(SURVEY.md 8d).  python tools/bench_ops_bf16.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import lib  # noqa: E402
from tools.bench_ops import CONVS, UPPROJ, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def plan(d):
    out = (C.c_int32 * 8)()
    lib().rd_gconv_bf16_plan_info(C.byref(d), out)
    return "(%d,%d) ckp%-2d %2dx%-3d lds %3dK wg %5d" % (out[0], out[1], out[2], out[3], out[4], out[6] // 1024, out[7])


def main():
    dev = "cuda"
    tot = 0.0
    rows = []
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device=dev)
        wp = ops.pack_weights_bf16(torch.randn(co, ci, k, k, device=dev))
        y = torch.empty(B, d.Ho, d.Wo, co, device=dev)
        rows.append((name, cnt, d, x, wp, y, 2.0 * B * d.Ho * d.Wo * co * ci * k * k))
    for name, c, h, w in UPPROJ:
        d = cd.upproj_fwd(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device=dev)
        wp = ops.pack_weights_bf16(torch.randn(c, c, 5, 5, device=dev))
        y = torch.empty(B, 2 * h, 2 * w, c, device=dev)
        rows.append((name, 1, d, x, wp, y, 2.0 * B * h * w * c * c * 25))
    for name, cnt, d, x, wp, y, flops in rows:
        t = timeit(lambda: ops.gconv_bf16(d, x, wp, y))
        byts = 4.0 * (x.numel() + y.numel())
        print("%-18s x%d %7.2f GF %7.1f MB | %7.1f us %6.1f TF %6.0f GB/s | %s" % (name, cnt, flops / 1e9, byts / 1e6, t * 1e6,
                                                                                flops / t / 1e12, byts / t / 1e9, plan(d)))
        tot += cnt * t
    print("TOTAL %.2f ms per forward (B=%d)" % (tot * 1e3, B))


if __name__ == "__main__":
    main()
