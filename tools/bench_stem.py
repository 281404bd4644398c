"""Timing of the 7x7/2 stem kernels (forward, weight gradient) at the bench geometry: python tools/bench_stem.py"""
import ctypes as C, sys, time
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from tools.bench_ops import timeit
L = lib()
N, H, W = 16, 450, 800
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
x = torch.rand(N, 4, H, W, device="cuda")
hw = H * W
for cin, cout, first in ((3, 64, 0), (1, 16, 3)):
    planes = (C.c_void_p * 3)(*[x.data_ptr() + 4 * hw * (first + c) if c < cin else None for c in range(3)])
    strides = (C.c_int64 * 3)(*[4 * hw if c < cin else 0 for c in range(3)])
    w = torch.randn(cout, cin, 7, 7, device="cuda") * 0.1
    wp = w.permute(2, 3, 1, 0).reshape(49, cin, cout).contiguous()
    out = torch.empty(N, Ho, Wo, cout, device="cuda")
    stat = torch.empty(L.rd_stem_stat_tiles(N, H, W), 2, cout, device="cuda")
    t = timeit(lambda: check(L.rd_stem_fwd(planes, strides, cin, N, H, W, ptr(wp), cout, ptr(out), ptr(stat), current_stream()), "stem_fwd"))
    gf = 2.0 * N * Ho * Wo * cout * 49 * cin / 1e9
    print("stem fwd  %d->%d: %7.1f us  %5.1f TF (%4.1f%% of fp32 peak)" % (cin, cout, t * 1e6, gf / t / 1e3, 100 * gf / t / 1e3 / 157.3))
    L.rd_stem_wgrad_workspace_floats.restype = C.c_int64
    ws = torch.empty(int(L.rd_stem_wgrad_workspace_floats(N, H, W, cin, cout)), device="cuda")
    gw = torch.empty(cout, cin, 7, 7, device="cuda")
    dout = torch.randn(N, Ho, Wo, cout, device="cuda")
    t = timeit(lambda: check(L.rd_stem_wgrad(planes, strides, cin, N, H, W, ptr(dout), cout, ptr(gw), ptr(ws), current_stream()), "stem_wgrad"))
    print("stem wgrad %d->%d: %7.1f us  %5.1f TF (%4.1f%% of fp32 peak)" % (cin, cout, t * 1e6, gf / t / 1e3, 100 * gf / t / 1e3 / 157.3))
    ts_ = timeit(lambda: check(L.rd_stem_wgrad_split_t(0, planes, strides, cin, N, H, W, ptr(dout), cout, ptr(gw), ptr(ws), current_stream()), "stem_wgrad_split"))
    d16 = dout.to(torch.bfloat16)
    tb_ = timeit(lambda: check(L.rd_stem_wgrad_split_t(1, planes, strides, cin, N, H, W, ptr(d16), cout, ptr(gw), ptr(ws), current_stream()), "stem_wgrad_split"))
    t16 = timeit(lambda: check(L.rd_stem_wgrad_t(1, planes, strides, cin, N, H, W, ptr(d16), cout, ptr(gw), ptr(ws), current_stream()), "stem_wgrad"))
    print("stem wgrad %d->%d on the bf16 matrix cores (three-piece operands): %7.1f us  %5.1f TF fp32-equivalent; bf16 dout: %7.1f us (fp32-MFMA kernel: %7.1f us)" % (cin, cout, ts_ * 1e6, gf / ts_ / 1e3, tb_ * 1e6, t16 * 1e6))
    # BatchNorm-backward apply pass of the stem (the kernel in front of the weight gradient at the end of the step)
    M = N * Ho * Wo
    raw = torch.randn(N, Ho, Wo, cout, device="cuda")
    gamma = torch.rand(cout, device="cuda") + 0.5
    mean = raw.mean((0, 1, 2)); invstd = 1.0 / torch.sqrt(raw.var((0, 1, 2), unbiased=False) + 1e-5)
    tiles = L.rd_bn_bwd_tiles(C.c_int64(M), cout)
    red = torch.zeros(tiles, 3, cout, device="cuda")
    check(L.rd_bn_bwd_reduce_t(0, ptr(dout), cout, None, 0, ptr(raw), cout, ptr(mean), None, 0, None, None, 0, C.c_int64(M), cout, 0, ptr(red), current_stream()), "reduce")
    dg, db, coef, dx = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda"), torch.zeros(3 * cout, device="cuda"), torch.empty_like(raw)
    ta = timeit(lambda: check(L.rd_bn_bwd_apply_t(0, ptr(dout), cout, ptr(raw), cout, ptr(red), tiles, 1, ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db), ptr(coef), ptr(dx), cout, C.c_int64(M), cout, current_stream()), "apply"))
    tf = timeit(lambda: check(L.rd_stem_wgrad_split_bn_t(0, planes, strides, cin, N, H, W, ptr(dout), ptr(raw), ptr(red), tiles, ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db), ptr(coef), cout, ptr(gw), ptr(ws), current_stream()), "fused"))
    r16, t16r = raw.to(torch.bfloat16), None
    dx16 = torch.empty_like(r16)
    ta16 = timeit(lambda: check(L.rd_bn_bwd_apply_t(1, ptr(d16), cout, ptr(r16), cout, ptr(red), tiles, 1, ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db), ptr(coef), ptr(dx16), cout, C.c_int64(M), cout, current_stream()), "apply"))
    tf16 = timeit(lambda: check(L.rd_stem_wgrad_split_bn_t(1, planes, strides, cin, N, H, W, ptr(d16), ptr(r16), ptr(red), tiles, ptr(gamma), ptr(mean), ptr(invstd), ptr(dg), ptr(db), ptr(coef), cout, ptr(gw), ptr(ws), current_stream()), "fused"))
    print("stem %d->%d: BatchNorm-backward apply %7.1f us + split weight gradient %7.1f us; folded into one launch %7.1f us | bf16 storage: %7.1f + %7.1f us; folded %7.1f us" % (cin, cout, ta * 1e6, ts_ * 1e6, tf * 1e6, ta16 * 1e6, tb_ * 1e6, tf16 * 1e6))
    # pooling + activation backward with the BatchNorm-backward sums (the first kernel of the stem's backward)
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.3
    pooled = torch.empty(N, Hp, Wp, cout, device="cuda"); idx = torch.empty(N, Hp, Wp, cout, dtype=torch.uint8, device="cuda")
    check(L.rd_bnact_maxpool_fwd(ptr(raw), ptr(sc), ptr(sh), 1, N, Ho, Wo, cout, ptr(pooled), cout, ptr(idx), current_stream()), "pool")
    dyp = torch.randn(N, Hp, Wp, cout, device="cuda")
    tl = L.rd_bnact_maxpool_bwd_tiles(N, Ho, Wo, cout)
    redp = torch.zeros(tl, 3, cout, device="cuda")
    tp = timeit(lambda: check(L.rd_bnact_maxpool_bwd_stats(ptr(dyp), cout, ptr(idx), ptr(raw), ptr(sc), ptr(sh), 1, N, Ho, Wo, cout, ptr(dx), ptr(mean), ptr(redp), current_stream()), "poolb"))
    tfw = timeit(lambda: check(L.rd_bnact_maxpool_fwd(ptr(raw), ptr(sc), ptr(sh), 1, N, Ho, Wo, cout, ptr(pooled), cout, ptr(idx), current_stream()), "pool"))
    mb = (raw.numel() * 8 + pooled.numel() * 5) / 1e6
    print("stem %d->%d: pool + act backward with sums %7.1f us (%.0f MB, %.2f TB/s); forward %7.1f us" % (cin, cout, tp * 1e6, mb, mb / tp / 1e6, tfw * 1e6))
