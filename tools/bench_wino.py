#!/usr/bin/env python3
"""Kernel-level gate of the Winograd F(2x2,3x3) split kernel: rd_wino_conv3x3 against the split kernels the plan runs
today (rd_gconv_split / rd_gconv_split_pre, whichever the planner prefers) and the fp32-MFMA kernel, per >= 64-channel 3x3 / stride-1 layer
at the bench geometry -- time, and error against an fp64 convolution (max |err| / max |out|) on a two-image slice.
    python tools/bench_wino.py [B]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from bench_ops import CONVS, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16


def err64(y_nhwc, x_nhwc, wt, nimg=2):
    ref = F.conv2d(x_nhwc[:nimg].permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
    return ((y_nhwc[:nimg].double().cpu() - ref).abs().max() / ref.abs().max()).item()


def main():
    dev = "cuda"
    tot = [0.0, 0.0, 0.0]
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        if k != 3 or s != 1 or not ops.wino_supported(h, w, ci, co):
            continue
        torch.manual_seed(1)
        x = torch.randn(B, h, w, ci, device=dev)
        wt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
        d = cd.conv_fwd(B, h, w, ci, co, 3, 1, 1)
        flops = 2.0 * B * h * w * co * ci * 9
        y32, ysp, yw = (torch.empty(B, h, w, co, device=dev) for _ in range(3))
        t32 = timeit(lambda: ops.gconv(d, x, ops.pack_weights(wt) if False else wp32, y32)) if (wp32 := ops.pack_weights(wt)) is not None else 0
        ws = ops.pack_weights_split(wt)
        tsp = timeit(lambda: ops.gconv_split(d, x, ws, ysp)) if ops.gconv_split_supported(d) else float("inf")
        tpre = float("inf")
        if ops.gconv_split_pre_supported(d):
            xp = ops.split_pieces(x)
            ypre = torch.empty_like(ysp)
            tpre = timeit(lambda: ops.gconv_split_pre(d, xp, ws, ypre))
            if tpre < tsp:
                ysp = ypre
        u = ops.wino_pack(wt)
        tw = timeit(lambda: ops.wino_conv3x3(x, u, yw))
        tpk = timeit(lambda: ops.wino_pack(wt))
        best = min(tsp, tpre)
        e32, esp, ew = err64(y32, x, wt), err64(ysp, x, wt), err64(yw, x, wt)
        print("%-18s x%d %6.2f GF | fp32 %6.1f us err %.1e | split %6.1f us (pre %6.1f) err %.1e | wino %6.1f us = %5.1f TF err %.1e (%.1fx fp32's) | x%.2f vs today | pack %5.1f us"
              % (name, cnt, flops / 1e9, t32 * 1e6, e32, tsp * 1e6, tpre * 1e6, esp, tw * 1e6, flops / tw / 1e12, ew, ew / e32, best / tw, tpk * 1e6), flush=True)
        for i, v in enumerate((t32, best, tw)):
            tot[i] += 2 * cnt * v          # forward + input gradient: the same kernel on the flipped operand
    print("TOTAL (forward + input gradient of these layers): fp32 MFMA %.2f ms | split today %.2f ms | winograd %.2f ms" % tuple(1e3 * v for v in tot))
    # dynamic range 2^-20 .. 2^20 and the input-gradient operand, on a small odd-sized shape
    for scale in (2.0 ** -20, 1.0, 2.0 ** 20):
        x = torch.randn(2, 29, 51, 64, device=dev) * scale
        wt = torch.randn(128, 64, 3, 3, device=dev) * 0.05
        y = torch.empty(2, 29, 51, 128, device=dev)
        ops.wino_conv3x3(x, ops.wino_pack(wt), y)
        dy = torch.randn(2, 29, 51, 128, device=dev) * scale
        dx = torch.empty(2, 29, 51, 64, device=dev)
        ops.wino_conv3x3(dy, ops.wino_pack(wt, flip=True), dx)
        ref_dx = F.conv_transpose2d(dy.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
        e_dx = ((dx.double().cpu() - ref_dx).abs().max() / ref_dx.abs().max()).item()
        print("scale 2^%+d: forward err %.2e, input-gradient err %.2e (odd 29 x 51, 64 -> 128)" % (round(torch.log2(torch.tensor(scale)).item()), err64(y, x, wt), e_dx))


if __name__ == "__main__":
    main()
