"""Race hunt over the PLANS of one descriptor: every candidate plan (rd_gconv_tune_candidates) is pinned in turn and launched REPS
times on the same inputs; all launches of a plan must agree bit for bit.   python tools/stress_plans.py N H W Cin Cout [reps]"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd
from radar_depth_amd._lib import check, current_stream, lib, ptr
L = lib()
L.rd_gconv_workspace_floats.restype = C.c_int64
n, h, w, ci, co = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 60
d = cd.conv_fwd(n, h, w, ci, co, 3, 1, 1)
cands = (C.c_int32 * (9 * 64))()
nc = L.rd_gconv_tune_candidates(C.byref(d), 1, cands, 64)
print(nc, "candidates")
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(n * h * w * ci, device="cuda", generator=g)
wp = torch.randn(9 * ci * co, device="cuda", generator=g)
out = torch.zeros(n * h * w * co, device="cuda")
for k in range(nc):
    cand = (C.c_int32 * 9)(*cands[9 * k:9 * k + 9])
    check(L.rd_gconv_tune_pin(C.byref(d), 1, cand), "pin")
    tiles = L.rd_gconv_stat_tiles_ws(C.byref(d))
    stat = torch.zeros(tiles * 2 * co, device="cuda")
    nws = L.rd_gconv_workspace_floats(C.byref(d))
    ws = torch.empty(max(int(nws), 1), device="cuda")
    ref, nbad, worst = None, 0, 0.0
    for it in range(reps):
        junk = torch.randn(1 << 20, device="cuda")
        out.fill_(float("nan"))
        check(L.rd_gconv_ws(C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, ptr(stat), ptr(ws), current_stream()), "gconv")
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(ref, out):
            nbad += 1
            dd = (ref - out).abs()
            worst = max(worst, float(dd.nan_to_num(1e9).max()))
            nel = int((dd > 0).sum())
    print("plan MT,NT,WM,WN,CKP,TH,TW,ksplit,pipe = %-40s %s" % (list(cand), "FLAKY %d/%d launches, %d elements, max |diff| %.3e" % (nbad, reps, nel, worst) if nbad else "ok"), flush=True)
