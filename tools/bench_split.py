#!/usr/bin/env python3
"""rd_gconv (fp32 MFMA) vs rd_gconv_split (six bf16 MFMAs per product) per layer at the bench geometry (B=16, 450x800): forward
and input gradient of every >= 32-channel conv shape of resnet18_latefusion.  TF = algorithmic fp32 FLOP / time.
Run on the GPU box:  python tools/bench_split.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import lib  # noqa: E402
from bench_ops import CONVS, UPPROJ, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16


def plan(d):
    v = (C.c_int32 * 8)()
    if lib().rd_gconv_split_plan_info(C.byref(d), v) != 0:
        return "-"
    return "%dx%d %dx%d lds %dK wg %d" % (v[0], v[1], v[2], v[3], v[5] // 1024, v[6])


def line(name, cnt, flops, t32, tsp, pl):
    print("%-22s x%d %7.2f GF | fp32 %8.1f us %6.1f TF | split %8.1f us %6.1f TF  x%.2f | %s"
          % (name, cnt, flops / 1e9, t32 * 1e6, flops / t32 / 1e12, tsp * 1e6, flops / tsp / 1e12, t32 / tsp, pl))


def main():
    dev = "cuda"
    tot32 = totsp = 0.0
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        if min(ci, co) < 32:
            continue
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        dd, zf = cd.conv_dgrad(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device=dev)
        wt = torch.randn(co, ci, k, k, device=dev)
        y = torch.empty(B, d.Ho, d.Wo, co, device=dev)
        dx = torch.zeros(B, h, w, ci, device=dev)
        flops = 2.0 * B * d.Ho * d.Wo * co * ci * k * k
        for tag, dsc, a, o, tr in (("fwd", d, x, y, False), ("dgrad", dd, y, dx, True)):
            wp = ops.pack_weights(wt, transpose=tr)
            t32 = timeit(lambda: ops.gconv(dsc, a, wp, o))
            if ops.gconv_split_supported(dsc):
                ws = ops.pack_weights_split(wt, transpose=tr)
                tsp = timeit(lambda: ops.gconv_split(dsc, a, ws, o))
            else:
                tsp = t32
            line(name + " " + tag, cnt, flops, t32, tsp, plan(dsc))
            tot32 += cnt * t32
            totsp += cnt * tsp
    for name, c, h, w in UPPROJ:
        if c < 32:
            continue
        d = cd.upproj_fwd(B, h, w, c, c)
        dd = cd.upproj_dgrad(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device=dev)
        wt = torch.randn(c, c, 5, 5, device=dev)
        y = torch.empty(B, 2 * h, 2 * w, c, device=dev)
        dx = torch.empty(B, h, w, c, device=dev)
        flops = 2.0 * B * h * w * c * c * 25
        for tag, dsc, a, o, tr in (("fwd", d, x, y, False), ("dgrad", dd, y, dx, True)):
            wp = ops.pack_weights(wt, transpose=tr)
            t32 = timeit(lambda: ops.gconv(dsc, a, wp, o))
            ws = ops.pack_weights_split(wt, transpose=tr)
            tsp = timeit(lambda: ops.gconv_split(dsc, a, ws, o)) if ops.gconv_split_supported(dsc) else t32
            line(name + " " + tag, 1, flops, t32, tsp, plan(dsc))
            tot32 += t32
            totsp += tsp
    print("TOTAL forward + input-gradient convolutions (>= 32 channels): fp32 MFMA %.2f ms, split %.2f ms" % (tot32 * 1e3, totsp * 1e3))


if __name__ == "__main__":
    main()
