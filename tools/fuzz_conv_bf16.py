"""Randomised parity sweep of the bf16-operand kernels (rd_gconv_bf16 forward + dgrad, rd_wgrad_bf16) against float64 torch
convolutions of the SAME bf16-rounded operands, over many small random geometries (tolerance 3e-5 of the max: fp32 summation
order only).   python tools/fuzz_conv_bf16.py [n_cases] [seed] [--poison]"""
import ctypes as C
import sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
POISON = "--poison" in sys.argv          # NaN-fill every CU's LDS before each launch: catches reads of unwritten LDS
sys.argv = [a for a in sys.argv if a != "--poison"]
if POISON:
    import os
    os.environ["RD_POISON_LDS"] = "1"
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def rel(a, b):
    return (a.double() - b).abs().max().item() / (b.abs().max().item() + 1e-30)


bad = 0
for case in range(n_cases):
    k, s = [(3, 1), (3, 1), (3, 2), (1, 1), (1, 2)][rng.randint(5)]
    p = k // 2
    ci = int(rng.choice([16, 32, 48, 64, 80, 96, 128, 160, 256, 320, 512]))
    co = int(rng.choice([16, 32, 48, 64, 96, 128, 192, 256]))
    n = int(rng.randint(1, 5))
    big = rng.rand() < 0.25
    h, w = (int(rng.randint(24, 130)), int(rng.randint(20, 210))) if big else (int(rng.randint(1, 70)), int(rng.randint(1, 90)))
    if big:
        ci, co = min(ci, 128), min(co, 128)
    g = torch.Generator().manual_seed(case)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (k * k * ci)) ** 0.5
    y = F.conv2d(bf(x), bf(wt), stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    tag = "n%d ci%d co%d k%d s%d %dx%d" % (n, ci, co, k, s, h, w)
    try:
        d = cd.conv_fwd(n, h, w, ci, co, k, s, p)
        xs, gys = ops.nchw_to_nhwc(x.cuda()), ops.nchw_to_nhwc(gy.cuda())
        out = torch.full((n, d.Ho, d.Wo, co), float("nan"), device="cuda")
        ops.gconv_bf16(d, xs, ops.pack_weights_bf16(wt.cuda()), out)
        e_f = rel(ops.nhwc_to_nchw(out).cpu(), y)
        dd, zero_fill = cd.conv_dgrad(n, h, w, ci, co, k, s, p)
        dx = torch.zeros(n, h, w, ci, device="cuda") if zero_fill else torch.full((n, h, w, ci), float("nan"), device="cuda")
        ops.gconv_bf16(dd, gys, ops.pack_weights_bf16(wt.cuda(), transpose=True), dx)
        ref_dx = torch.nn.grad.conv2d_input(x.shape, bf(wt), bf(gy), stride=s, padding=p)
        e_d = rel(ops.nhwc_to_nchw(dx).cpu(), ref_dx)
        e_w = 0.0
        if lib().rd_wgrad_bf16_supported(C.byref(d)) == 1:
            gw = torch.full((co, ci, k, k), float("nan"), device="cuda")
            ops.wgrad_bf16(d, xs, gys, gw)
            e_w = rel(gw.cpu(), torch.nn.grad.conv2d_weight(bf(x), wt.shape, bf(gy), stride=s, padding=p))
        ok = max(e_f, e_d, e_w) < 3e-5 and e_f == e_f and e_d == e_d and e_w == e_w
        if not ok:
            bad += 1
            print("MISMATCH", tag, "fwd %.2e dgrad %.2e wgrad %.2e" % (e_f, e_d, e_w))
    except Exception as e:   # noqa: BLE001
        bad += 1
        print("ERROR", tag, str(e)[:120])
print("%d cases, %d bad%s" % (n_cases, bad, " (LDS poisoned)" if POISON else ""))
sys.exit(1 if bad else 0)
