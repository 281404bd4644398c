#!/usr/bin/env python3
"""Tile sweep of rd_gconv_bf16 for chosen layers of tools/bench_ops.py (planner diagnostics).  The planner reads its overrides
once per process, so every (layer, tile, CKP, pipe) point runs in its own interpreter:
    python tools/sweep_gconv_bf16.py "layer4 3x3 512" "dec4 c2 3x3 16"     (no arguments: all)
    python tools/sweep_gconv_bf16.py --one <index>                        (child mode)"""
import os, subprocess, sys
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    import torch
    from radar_depth_amd import convdesc as cd, ops
    from tools.bench_ops import CONVS, UPPROJ, timeit
    from tools.bench_ops_bf16 import plan
    i, B = int(sys.argv[2]), 16
    if i < len(CONVS):
        name, cnt, ci, co, k, s, p, h, w = CONVS[i]
        d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
        x = torch.randn(B, h, w, ci, device="cuda")
        wp = ops.pack_weights_bf16(torch.randn(co, ci, k, k, device="cuda"))
        y = torch.empty(B, d.Ho, d.Wo, co, device="cuda")
    else:
        name, c, h, w = UPPROJ[i - len(CONVS)]
        d = cd.upproj_fwd(B, h, w, c, c)
        x = torch.randn(B, h, w, c, device="cuda")
        wp = ops.pack_weights_bf16(torch.randn(c, c, 5, 5, device="cuda"))
        y = torch.empty(B, 2 * h, 2 * w, c, device="cuda")
    IO16 = os.environ.get("RD_SWEEP_IO16") == "1"        # bf16-storage form of the kernel
    if IO16:
        import ctypes as C
        from radar_depth_amd._lib import current_stream, lib, ptr
        x, y = x.to(torch.bfloat16), y.to(torch.bfloat16)
        run = lambda: lib().rd_gconv_bf16_t(1, C.byref(d), ptr(x), ptr(wp), ptr(y), None, 0, 0, None, 0, None, current_stream())
    else:
        run = lambda: ops.gconv_bf16(d, x, wp, y)
    try:
        t = timeit(run)
        print("%-18s %8.1f us  %s" % (name, t * 1e6, plan(d)))
    except Exception as e:
        print("%-18s infeasible (%s)" % (name, str(e)[:60]))
    sys.exit(0)
from tools.bench_ops import CONVS, UPPROJ
names = [c[0] for c in CONVS] + [u[0] for u in UPPROJ]
want = sys.argv[1:] or names
for i, n in enumerate(names):
    if n not in want:
        continue
    seen = set()
    for force in range(5):
        for ckp in (16, 32, 64):
            for nopipe in (0, 1):
                env = dict(os.environ, RD_GCONV_BF16_FORCE=str(force), RD_GCONV_BF16_CKP=str(ckp))
                if nopipe:
                    env["RD_GCONV_BF16_NOPIPE"] = "1"
                r = subprocess.run([sys.executable, __file__, "--one", str(i)], env=env, capture_output=True, text=True)
                out = r.stdout.strip().splitlines()
                line = out[-1] if out else "? " + r.stderr.strip().splitlines()[-1][:100]
                key = line.split("us", 1)[-1]
                if key in seen:
                    continue
                seen.add(key)
                print(line, flush=True)
