#!/usr/bin/env python3
"""rd_gconv (fp32 MFMA) vs rd_gconv_split (split while staging) vs rd_gconv_split_pre (activation split by its producer: staging =
global_load_lds only) per layer at the bench geometry (B=16, 450x800), forward and input gradient of every >= 64-channel 3x3 / 5x5
shape; plus the stand-alone split pass.  TF = algorithmic fp32 FLOP / time.   python tools/bench_split_pre.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import lib  # noqa: E402
from bench_ops import CONVS, UPPROJ, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16


def plan(d, pre):
    v = (C.c_int32 * 8)()
    f = lib().rd_gconv_split_pre_plan_info if pre else lib().rd_gconv_split_plan_info
    if f(C.byref(d), v) != 0:
        return "-"
    return "%dx%d %dx%d lds %dK wg %d" % (v[0], v[1], v[2], v[3], v[5] // 1024, v[6])


tot = [0.0, 0.0, 0.0, 0.0]


def one(name, cnt, flops, dsc, a, o, wt, tr):
    wp = ops.pack_weights(wt, transpose=tr)
    t32 = timeit(lambda: ops.gconv(dsc, a, wp, o))
    tsp = tpre = t32
    tpc = 0.0
    if ops.gconv_split_supported(dsc) or ops.gconv_split_pre_supported(dsc):
        ws = ops.pack_weights_split(wt, transpose=tr)
        if ops.gconv_split_supported(dsc):
            tsp = timeit(lambda: ops.gconv_split(dsc, a, ws, o))
        if ops.gconv_split_pre_supported(dsc):
            ap = ops.split_pieces(a)
            tpre = timeit(lambda: ops.gconv_split_pre(dsc, ap, ws, o))
            tpc = timeit(lambda: ops.split_pieces(a))
    print("%-22s x%d %7.2f GF | fp32 %7.1f us %5.1f TF | split %7.1f us %5.1f TF | pre %7.1f us %5.1f TF  x%.2f vs split | split pass %6.1f us | %s | %s"
          % (name, cnt, flops / 1e9, t32 * 1e6, flops / t32 / 1e12, tsp * 1e6, flops / tsp / 1e12, tpre * 1e6, flops / tpre / 1e12, tsp / tpre,
             tpc * 1e6, plan(dsc, False), plan(dsc, True)), flush=True)
    for i, v in enumerate((t32, tsp, tpre, tpc)):
        tot[i] += cnt * v


def main():
    dev = "cuda"
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        if min(ci, co) < 32:
            continue
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        dd, zf = cd.conv_dgrad(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device=dev)
        wt = torch.randn(co, ci, k, k, device=dev)
        y = torch.randn(B, d.Ho, d.Wo, co, device=dev)
        dx = torch.zeros(B, h, w, ci, device=dev)
        flops = 2.0 * B * d.Ho * d.Wo * co * ci * k * k
        one(name + " fwd", cnt, flops, d, x, torch.empty_like(y), wt, False)
        one(name + " dgrad", cnt, flops, dd, y, dx, wt, True)
    for name, c, h, w in UPPROJ:
        if c < 32:
            continue
        d = cd.upproj_fwd(B, h, w, c, c)
        dd = cd.upproj_dgrad(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device=dev)
        wt = torch.randn(c, c, 5, 5, device=dev)
        y = torch.randn(B, 2 * h, 2 * w, c, device=dev)
        flops = 2.0 * B * h * w * c * c * 25
        one(name + " fwd", 1, flops, d, x, torch.empty_like(y), wt, False)
        one(name + " dgrad", 1, flops, dd, y, torch.empty(B, h, w, c, device=dev), wt, True)
    print("TOTAL forward + input-gradient convolutions: fp32 MFMA %.2f ms | split %.2f ms | pre-split %.2f ms (+ %.2f ms if every operand were split by a stand-alone pass)"
          % (tot[0] * 1e3, tot[1] * 1e3, tot[2] * 1e3, tot[3] * 1e3))


if __name__ == "__main__":
    main()
