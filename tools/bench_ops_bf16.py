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

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16


def plan(d):
    out = (C.c_int32 * 8)()
    lib().rd_gconv_bf16_plan_info(C.byref(d), out)
    return "(%d,%d)%s ckp%-2d %2dx%-3d lds %3dK wg %5d" % (out[0], out[1], "P" if out[2] >= 1000 else " ", out[2] % 1000, out[3], out[4],
                                                          out[6] // 1024, out[7])


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
    totw = 0.0
    for name, cnt, d, x, wp, y, flops in rows:
        t = timeit(lambda: ops.gconv_bf16(d, x, wp, y))
        byts = 4.0 * (x.numel() + y.numel())
        wg = ""
        if lib().rd_wgrad_bf16_supported(C.byref(d)) == 1:
            n = int(lib().rd_wgrad_bf16_workspace_floats(C.byref(d)))
            slabs = torch.empty(n, device=dev)
            kk = int(round(wp.shape[0] ** 0.5))          # 3, 1 or 5 (UpProj)
            grad = torch.empty(d.Cout, d.Cin, kk, kk, device=dev)
            cs = ops.current_stream

            def wgf():
                assert lib().rd_wgrad_bf16(C.byref(d), ops.ptr(x), ops.ptr(y), ops.ptr(slabs), cs()) == 0
                assert lib().rd_wgrad_bf16_reduce(C.byref(d), ops.ptr(slabs), ops.ptr(grad), d.Cout, d.Cin, kk, kk, 0, 0, cs()) == 0
            tw = timeit(wgf)
            info = (C.c_int32 * 6)()
            lib().rd_wgrad_bf16_plan_info(C.byref(d), info)
            wg = " | wgrad %7.1f us %6.1f TF %6.0f GB/s blocks %d splits %d slabs %d" % (tw * 1e6, flops / tw / 1e12, byts / tw / 1e9, info[2], info[3], info[4])
            totw += cnt * tw
        print("%-18s x%d %7.2f GF %7.1f MB | %7.1f us %6.1f TF %6.0f GB/s | %s%s" % (name, cnt, flops / 1e9, byts / 1e6, t * 1e6,
                                                                                  flops / t / 1e12, byts / t / 1e9, plan(d), wg))
        tot += cnt * t
    print("TOTAL %.2f ms per forward (B=%d); bf16 weight gradients (incl. slab reduction) %.2f ms" % (tot * 1e3, B, totw * 1e3))


if __name__ == "__main__":
    main()
