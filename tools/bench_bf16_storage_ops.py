#!/usr/bin/env python3
"""Per-layer table of the bf16-STORAGE convolution (rd_gconv_bf16_t with bf16 tensors: BASELINE configs 3 / 5) at the bench geometry
(B=16, 450x800): forward and input gradient of every conv shape, time, TFLOP/s, algorithmic GB/s (bf16 input read once + bf16 output
written once), and the plan (P = the persistent pipelined kernel of csrc/gconv_bf16p.hip).  RD_GCONV_BF16P=0 puts every shape on
gconv_bf16_kernel (the A/B: run the tool twice).      python tools/bench_bf16_storage_ops.py [B]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import check, current_stream, lib, ptr  # noqa: E402
from bench_ops import CONVS, UPPROJ, timeit  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
BF16 = 1
L = lib()
tot = {"P": 0.0, "-": 0.0}


def plan(d):
    v = (C.c_int32 * 8)()
    if L.rd_gconv_bf16_plan_info_t(BF16, C.byref(d), v) != 0:
        return "?", "-"
    pk = v[2] >= 2000
    return "%s(%d,%d) %2dx%-3d lds %3dK jobs %5d" % ("P" if pk else " ", v[0], v[1], v[3], v[4], v[6] // 1024, v[7]), "P" if pk else "-"


def one(name, cnt, flops, d, x, wt, out, tr):
    wp = ops.pack_weights_bf16(wt, transpose=tr)
    t = timeit(lambda: check(L.rd_gconv_bf16_t(BF16, C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, 0, None, 0, None, current_stream()), "gconv_bf16_t"))
    byts = 2.0 * (x.numel() + out.numel())
    ps, kind = plan(d)
    tot[kind] += cnt * t
    print("%-22s x%d %7.2f GF %7.1f MB | %7.1f us %6.1f TF %6.0f GB/s | %s" % (name, cnt, flops / 1e9, byts / 1e6, t * 1e6, flops / t / 1e12, byts / t / 1e9, ps), flush=True)


def main():
    dev = "cuda"
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        if ci % 16 or co % 16:
            continue
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        dd, zf = cd.conv_dgrad(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device=dev).to(torch.bfloat16)
        wt = torch.randn(co, ci, k, k, device=dev)
        y = torch.randn(B, d.Ho, d.Wo, co, device=dev).to(torch.bfloat16)
        flops = 2.0 * B * d.Ho * d.Wo * co * ci * k * k
        one(name + " fwd", cnt, flops, d, x, wt, torch.empty_like(y), False)
        one(name + " dgrad", cnt, flops, dd, y, wt, torch.zeros_like(x), True)
    for name, c, h, w in UPPROJ:
        d = cd.upproj_fwd(B, h, w, c, c)
        dd = cd.upproj_dgrad(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device=dev).to(torch.bfloat16)
        wt = torch.randn(c, c, 5, 5, device=dev)
        y = torch.randn(B, 2 * h, 2 * w, c, device=dev).to(torch.bfloat16)
        flops = 2.0 * B * h * w * c * c * 25
        one(name + " fwd", 1, flops, d, x, wt, torch.empty_like(y), False)
        one(name + " dgrad", 1, flops, dd, y, wt, torch.empty_like(x), True)
    print("TOTAL forward + input-gradient convolutions of one step: %.3f ms on the persistent kernel, %.3f ms on gconv_bf16_kernel" % (tot["P"] * 1e3, tot["-"] * 1e3))


if __name__ == "__main__":
    main()
