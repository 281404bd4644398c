import ctypes as C, os, sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import check, current_stream, lib, ptr
from bench_ops import timeit
L = lib(); B = 16
for name, (ci, co, h, w) in {"layer1": (64, 64, 113, 200), "layer2": (128, 128, 57, 100)}.items():
    d = cd.conv_fwd(B, h, w, ci, co, 3, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda").to(torch.bfloat16)
    wp = ops.pack_weights_bf16(torch.randn(co, ci, 3, 3, device="cuda"))
    y = torch.empty(B, h, w, co, device="cuda", dtype=torch.bfloat16)
    row = []
    for grid in (128, 256, 384, 512, 768, 1024):
        os.environ["RD_GCONV_BF16P_GRID"] = str(grid)
        for dbg in (0, 16):
            os.environ["RD_GCONV_BF16P_DEBUG"] = str(dbg)
            t = timeit(lambda: check(L.rd_gconv_bf16_t(1, C.byref(d), ptr(x), ptr(wp), ptr(y), None, 0, 0, None, 0, None, current_stream()), "g"))
            row.append("grid%d%s %.1f" % (grid, "/noEPI" if dbg else "", t * 1e6))
    print(name, " | ".join(row), flush=True)
