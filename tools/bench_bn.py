"""Streaming rate of the BatchNorm-family kernels at the bench geometry (fp32 or bf16 storage): algorithmic bytes / time.
    python tools/bench_bn.py [fp32|bf16]"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from tools.bench_ops import timeit
L = lib()
MODE = sys.argv[1] if len(sys.argv) > 1 else "fp32"
DT, adt, esz = (1, torch.bfloat16, 2) if MODE == "bf16" else (0, torch.float32, 4)
SHAPES = [("layer1 64ch 113x200", 16 * 113 * 200, 64), ("layer2 128ch 57x100", 16 * 57 * 100, 128), ("layer3 256ch 29x50", 16 * 29 * 50, 256),
          ("layer4 512ch 15x25", 16 * 15 * 25, 512), ("dec3 32ch 120x200", 16 * 120 * 200, 32), ("dec4 16ch 240x400", 16 * 240 * 400, 16),
          ("depth1 16ch 113x200", 16 * 113 * 200, 16)]
for name, M, Cc in SHAPES:
    x = torch.randn(M, Cc, device="cuda").to(adt)
    x2 = torch.randn(M, Cc, device="cuda").to(adt)
    dy = torch.randn(M, Cc, device="cuda").to(adt)
    y = torch.empty(M, Cc, device="cuda", dtype=adt)
    g = torch.empty(M, Cc, device="cuda", dtype=adt)
    dx = torch.empty(M, Cc, device="cuda", dtype=adt)
    sc, sh, mu, inv, gam = (torch.rand(Cc, device="cuda") + 0.5 for _ in range(5))
    dg, db, coef = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda"), torch.empty(6 * Cc, device="cuda")
    tiles = L.rd_bn_bwd_tiles(C.c_int64(M), Cc)
    red = torch.empty(tiles, 3, Cc, device="cuda")
    s = current_stream()
    T = M * Cc * esz / 1e6      # MB per tensor pass
    rows = []
    t = timeit(lambda: check(L.rd_bn_act_t(DT, ptr(x), Cc, ptr(sc), ptr(sh), None, 0, None, None, ptr(y), Cc, C.c_int64(M), Cc, 1, s), "act"))
    rows.append(("bn_act lone (2 passes)", t, 2 * T))
    t = timeit(lambda: check(L.rd_bn_act_t(DT, ptr(x), Cc, ptr(sc), ptr(sh), ptr(x2), Cc, None, None, ptr(y), Cc, C.c_int64(M), Cc, 1, s), "act2"))
    rows.append(("bn_act + residual (3)", t, 3 * T))
    t = timeit(lambda: check(L.rd_bn_bwd_reduce_x_t(DT, ptr(dy), Cc, ptr(x), Cc, ptr(mu), ptr(sc), ptr(sh), None, 0, C.c_int64(M), Cc, 1, ptr(red), s), "redx"))
    rows.append(("bwd_reduce lone (2)", t, 2 * T))
    t = timeit(lambda: check(L.rd_bn_bwd_reduce_t(DT, ptr(dy), Cc, ptr(y), Cc, ptr(x), Cc, ptr(mu), None, 0, None, ptr(g), Cc, C.c_int64(M), Cc, 1, ptr(red), s), "red"))
    rows.append(("bwd_reduce join (4)", t, 4 * T))
    t = timeit(lambda: check(L.rd_bn_bwd_apply_x_t(DT, ptr(dy), Cc, ptr(x), Cc, ptr(red), tiles, ptr(gam), ptr(mu), ptr(inv), ptr(sc), ptr(sh), 1, ptr(dg), ptr(db), ptr(coef),
                                                   ptr(dx), Cc, C.c_int64(M), Cc, s), "applyx"))
    rows.append(("bwd_apply lone (3) + coeffs", t, 3 * T))
    print("%-22s %6.1f MB/pass | " % (name, T) + " | ".join("%s %6.1f us %4.2f TB/s" % (n, tt * 1e6, mb / tt / 1e6) for n, tt, mb in rows))
