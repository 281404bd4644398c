"""Race hunt over the PLANS of one descriptor: every candidate plan (rd_gconv_tune_candidates) is pinned in turn and launched REPS
times on the same inputs; all launches of a plan must agree bit for bit, and every plan must agree with the first one to 1e-4.
   python tools/stress_plans.py N H W Cin Cout [reps] [stride | up | updgrad]     (stride 2: the input-parity-group kernels;
   up / updgrad: the 4-phase UpProj forward / its 25-tap input gradient on an HxW low-resolution map)"""
import ctypes as C, sys
import torch
sys.path.insert(0, ".")
from radar_depth_amd import convdesc as cd
from radar_depth_amd._lib import check, current_stream, lib, ptr
L = lib()
L.rd_gconv_workspace_floats.restype = C.c_int64
n, h, w, ci, co = (int(v) for v in sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 60
mode = sys.argv[7] if len(sys.argv) > 7 else "1"
if mode == "up":
    d = cd.upproj_fwd(n, h, w, ci, co)
elif mode == "updgrad":
    d = cd.upproj_dgrad(n, h, w, ci, co)
else:
    d = cd.conv_fwd(n, h, w, ci, co, 3, int(mode), 1)
n_slabs = max(d.phase[i].widx[t] for i in range(d.n_phases) for t in range(d.phase[i].n_taps)) + 1
cands = (C.c_int32 * (9 * 64))()
nc = L.rd_gconv_tune_candidates(C.byref(d), 1, cands, 64)
print(nc, "candidates")
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(d.N * d.Hi * d.Wi * d.ldi, device="cuda", generator=g)
wp = torch.randn(n_slabs * d.Cin * d.Cout, device="cuda", generator=g)
out = torch.zeros(d.N * d.Ho * d.Wo * d.ldo, device="cuda")
co = d.Cout
first = None
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
    if first is None:
        first = ref.clone()
    dev_ = float(((ref - first).abs().max() / first.abs().max()).nan_to_num(1e9))
    print("plan MT,NT,WM,WN,CKP,TH,TW,ksplit,pipe = %-40s %s%s" % (list(cand), "FLAKY %d/%d launches, %d elements, max |diff| %.3e" % (nbad, reps, nel, worst) if nbad else "ok",
                                                                    "" if dev_ < 1e-4 else "  WRONG vs first plan: %.3e" % dev_), flush=True)
