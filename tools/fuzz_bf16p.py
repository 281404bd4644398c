#!/usr/bin/env python3
"""Randomised sweep of the persistent bf16-storage convolution (csrc/gconv_bf16p.hip) against fp64 convolutions of the same bf16 operands:
3x3 forward (with BatchNorm partial sums), its input gradient with a residual addend, the four-phase UpProj forward; random batch, channel
counts, image sizes, persistent-grid caps (1 .. the launch's own grid) and, every third case, NaN-poisoned LDS.  Every output must be within
one bf16 ulp, the statistics within 1e-4.      python tools/fuzz_bf16p.py [cases] [seed]"""
import ctypes as C
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import check, current_stream, lib, ptr  # noqa: E402

L = lib()
BF16, ULP = 1, 2.0 ** -8
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
L.rd_gconv_bf16p_plan_all(1)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def ulp_err(got, ref):
    scale = torch.maximum(ref.abs(), torch.full_like(ref, 1e-3 * ref.abs().max().item()))
    return ((got.double().cpu() - ref).abs() / (scale * ULP)).max().item()


def nhwc16(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


bad = launches = 0
for case in range(cases):
    kind = rng.choice(["conv", "conv", "upproj"])
    n = rng.randint(1, 4)
    ci = 16 * rng.randint(2, 10)
    co = 32 * rng.randint(1, 6)
    h, w = rng.randint(1, 40), rng.randint(1, 70)
    if kind == "upproj":
        h, w = max(1, h // 2), max(1, w // 2)
    cap = rng.choice([0, 1, 2, 7, 64])
    poison = case % 3 == 2
    if cap:
        os.environ["RD_GCONV_BF16P_GRID"] = str(cap)
    else:
        os.environ.pop("RD_GCONV_BF16P_GRID", None)
    g = torch.Generator().manual_seed(1000 + case)
    x = bf(torch.randn(n, ci, h, w, generator=g))
    if kind == "upproj":
        wt = torch.randn(co, ci, 5, 5, generator=g) * (2.0 / (9 * ci)) ** 0.5
        up = torch.zeros(n, ci, 2 * h, 2 * w)
        up[:, :, ::2, ::2] = x
        ref = F.conv2d(up.double(), bf(wt).double(), None, 1, 2)
        d = cd.upproj_fwd(n, h, w, ci, co)
    else:
        wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5
        ref = F.conv2d(x.double(), bf(wt).double(), None, 1, 1)
        d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
    info = (C.c_int32 * 8)()
    if L.rd_gconv_bf16_plan_info_t(BF16, C.byref(d), info) != 0 or info[2] < 2000:
        continue
    xs, wp = nhwc16(x), ops.pack_weights_bf16(wt.cuda())
    out = torch.full((n, d.Ho, d.Wo, co), float("nan"), dtype=torch.bfloat16, device="cuda")
    stat = torch.full((L.rd_gconv_bf16_stat_tiles_t(BF16, C.byref(d)), 2, co), float("nan"), device="cuda")
    if poison:
        check(L.rd_debug_poison_lds(current_stream()), "poison")
    check(L.rd_gconv_bf16_t(BF16, C.byref(d), ptr(xs), ptr(wp), ptr(out), None, 0, 0, None, 0, ptr(stat), current_stream()), "fwd")
    torch.cuda.synchronize()
    launches += 1
    e = ulp_err(out.permute(0, 3, 1, 2), ref)
    s_ = stat.double().sum(0).cpu()
    es = ((s_[0] - ref.sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).sqrt().max().clamp_min(1e-30)).item()
    eq = ((s_[1] - (ref ** 2).sum((0, 2, 3))).abs().max() / (ref ** 2).sum((0, 2, 3)).max().clamp_min(1e-30)).item()
    ok = e <= 1.01 and es < 1e-4 and eq < 1e-4
    ed = 0.0
    if kind == "conv" and ci % 32 == 0:
        gy = bf(torch.randn(ref.shape, generator=g))
        add = bf(torch.randn(n, ci, h, w, generator=g))
        dref = torch.nn.grad.conv2d_input((n, ci, h, w), bf(wt).double(), gy.double(), 1, 1) + add.double()
        dd, _ = cd.conv_dgrad(n, h, w, ci, co, 3, 1, 1)
        if L.rd_gconv_bf16_plan_info_t(BF16, C.byref(dd), info) == 0 and info[2] >= 2000:
            dx = torch.full((n, h, w, ci), float("nan"), dtype=torch.bfloat16, device="cuda")
            wd = ops.pack_weights_bf16(wt.cuda(), transpose=True)
            adds, gys = nhwc16(add), nhwc16(gy)
            check(L.rd_gconv_bf16_t(BF16, C.byref(dd), ptr(gys), ptr(wd), ptr(dx), None, 0, 0, ptr(adds), ci, None, current_stream()), "dgrad")
            torch.cuda.synchronize()
            launches += 1
            ed = ulp_err(dx.permute(0, 3, 1, 2), dref)
            ok = ok and ed <= 1.01
    if not ok:
        bad += 1
        print("FAIL case %d %s n=%d ci=%d co=%d %dx%d cap=%d poison=%d: out %.2f ulp, stat %.1e / %.1e, dgrad %.2f ulp" % (case, kind, n, ci, co, h, w, cap, poison, e, es, eq, ed), flush=True)
print("%d cases, %d kernel launches on gconv_bf16p_kernel, %d failures" % (cases, launches, bad))
