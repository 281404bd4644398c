#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv kernels at the bench geometry (B=16, 450x800): time, TFLOP/s
and fraction of the fp32 peak (157.3 TFLOP/s) for forward, dgrad and wgrad of every conv shape of
resnet18_latefusion.  Run on the GPU box:  python tools/bench_ops.py [B]"""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402

PEAK = 157.3
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
# (name, count, Cin, Cout, k, s, p, Hin, Win)
CONVS = [
    ("layer1 3x3 64", 4, 64, 64, 3, 1, 1, 113, 200),
    ("layer2.0.c1 s2", 1, 64, 128, 3, 2, 1, 113, 200),
    ("layer2 3x3 128", 3, 128, 128, 3, 1, 1, 57, 100),
    ("layer2 ds 1x1", 1, 64, 128, 1, 2, 0, 113, 200),
    ("layer3.0.c1 s2", 1, 128, 256, 3, 2, 1, 57, 100),
    ("layer3 3x3 256", 3, 256, 256, 3, 1, 1, 29, 50),
    ("layer4.0.c1 s2", 1, 256, 512, 3, 2, 1, 29, 50),
    ("layer4 3x3 512", 3, 512, 512, 3, 1, 1, 15, 25),
    ("d.layer1 3x3 16", 4, 16, 16, 3, 1, 1, 113, 200),
    ("d.layer2 3x3 32", 3, 32, 32, 3, 1, 1, 57, 100),
    ("d.layer3 3x3 64", 3, 64, 64, 3, 1, 1, 29, 50),
    ("d.layer4 3x3 128", 3, 128, 128, 3, 1, 1, 15, 25),
    ("fusion 1x1 640", 1, 640, 512, 1, 1, 0, 15, 25),
    ("conv2 1x1 512", 1, 512, 256, 1, 1, 0, 15, 25),
    ("dec1 c2 3x3 128", 1, 128, 128, 3, 1, 1, 30, 50),
    ("dec2 c2 3x3 64", 1, 64, 64, 3, 1, 1, 60, 100),
    ("dec3 c2 3x3 32", 1, 32, 32, 3, 1, 1, 120, 200),
    ("dec4 c2 3x3 16", 1, 16, 16, 3, 1, 1, 240, 400),
]
UPPROJ = [("dec1 up5x5 256", 256, 15, 25), ("dec2 up5x5 128", 128, 30, 50), ("dec3 up5x5 64", 64, 60, 100),
          ("dec4 up5x5 32", 32, 120, 200)]


def timeit(fn, iters=10):
    """Seconds per call.  The device clock ramps from ~2.06 GHz to ~2.38 GHz over tens of milliseconds of continuous load (an
    isolated handful of launches after an idle gap under-reports by ~13 %), so the op is first run back to back for >= 60 ms."""
    fn()
    torch.cuda.synchronize()
    w0 = time.perf_counter()
    while time.perf_counter() - w0 < 0.06:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e-3


def report(name, cnt, flops, tf, td, tw):
    def f(t):
        return "%8.1f us %6.1f TF %4.0f%%" % (t * 1e6, flops / t / 1e12, 100 * flops / t / 1e12 / PEAK) if t else " " * 30
    print("%-18s x%d %7.2f GF | fwd %s | dgrad %s | wgrad %s" % (name, cnt, flops / 1e9, f(tf), f(td), f(tw)))
    return cnt * (tf + (td or 0) + tw), cnt * flops * (3 if td else 2)


def main():
    dev = "cuda"
    tot_t = tot_f = 0.0
    for name, cnt, ci, co, k, s, p, h, w in CONVS:
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device=dev)
        wt = torch.randn(co, ci, k, k, device=dev)
        wp = ops.pack_weights(wt)
        wd = ops.pack_weights(wt, transpose=True)
        y = torch.empty(B, d.Ho, d.Wo, co, device=dev)
        stat = torch.zeros(ops.gconv_stat_tiles(d), 2, co, device=dev)
        dd, zf = cd.conv_dgrad(B, h, w, ci, co, k, s, p)
        dx = torch.zeros(B, h, w, ci, device=dev)
        slabs = torch.empty(ops.wgrad_workspace_floats(d), device=dev)
        grad = torch.empty_like(wt)
        flops = 2.0 * B * d.Ho * d.Wo * co * ci * k * k
        tf = timeit(lambda: ops.gconv(d, x, wp, y, stat=stat))
        td = timeit(lambda: ops.gconv(dd, y, wd, dx))

        def wg():
            ops.wgrad(d, x, y, slabs)
            ops.wgrad_reduce(d, slabs, grad)
        tw = timeit(wg)
        t, f = report(name, cnt, flops, tf, td, tw)
        tot_t += t
        tot_f += f
    for name, c, h, w in UPPROJ:
        d = cd.upproj_fwd(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device=dev)
        wp = torch.randn(25, c, c, device=dev)
        y = torch.empty(B, 2 * h, 2 * w, c, device=dev)
        stat = torch.zeros(ops.gconv_stat_tiles(d), 2, c, device=dev)
        dd = cd.upproj_dgrad(B, h, w, c, c)
        dx = torch.empty(B, h, w, c, device=dev)
        slabs = torch.empty(ops.wgrad_workspace_floats(d), device=dev)
        grad = torch.empty(c // 2, c, 5, 5, device=dev)
        flops = 2.0 * B * h * w * c * c * 25
        tf = timeit(lambda: ops.gconv(d, x, wp, y, stat=stat))
        td = timeit(lambda: ops.gconv(dd, y, wp, dx))

        def wg():
            ops.wgrad(d, x, y, slabs)
            ops.wgrad_reduce(d, slabs, grad, co_off=0)
            ops.wgrad_reduce(d, slabs, grad, co_off=c // 2)
        tw = timeit(wg)
        t, f = report(name, 1, flops, tf, td, tw)
        tot_t += t
        tot_f += f
    print("TOTAL conv time %.2f ms for %.1f GFLOP -> %.1f TF (%.0f%% of fp32 peak); => <= %.0f samples/s from convs alone"
          % (tot_t * 1e3, tot_f / 1e9, tot_f / tot_t / 1e12, 100 * tot_f / tot_t / 1e12 / PEAK, B / tot_t))


if __name__ == "__main__":
    main()
