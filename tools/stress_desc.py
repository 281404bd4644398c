"""Race hunt at the level of single launches: every fp32 gconv descriptor of a model's plans (forward and dgrad) is launched REPS
times on the same inputs; outputs (and BatchNorm partial sums) must be bit-identical every time.
    python tools/stress_desc.py [arch] [batch] [height] [width] [reps]"""
import ctypes as C, sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import procedural_fill_

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18_multistage_uncertainty_fixs"
b, h, w = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 450, 800)))
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 25
L = lib()
L.rd_gconv_workspace_floats.restype = C.c_int64
args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
made = create_model(args, [h, w])
m, lw = made if isinstance(made, tuple) else (made, None)
procedural_fill_(m)
ts = HipTrainStep(m.cuda(), b, h, w, loss_weights=lw)
plans = [ts.plan] if hasattr(ts, "plan") and ts.plan is not None else []
if hasattr(ts, "mp"):
    plans = [ts.mp.p1, ts.mp.p2]
seen, bad = set(), 0
for pl in plans:
    for name, (kind, d) in pl.meta.items():
        if kind != "gconv":
            continue
        key = bytes(d)
        if key in seen:
            continue
        seen.add(key)
        S = max(d.phase[i].widx[t] for i in range(d.n_phases) for t in range(d.phase[i].n_taps)) + 1
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(d.N * d.Hi * d.Wi * d.ldi, device="cuda", generator=g)
        wp = torch.randn(S * d.Cin * d.Cout, device="cuda", generator=g)
        out = torch.zeros(d.N * d.Ho * d.Wo * d.ldo, device="cuda")
        add = torch.randn(d.N * d.Ho * d.Wo * d.ldo, device="cuda", generator=g)
        tiles = L.rd_gconv_stat_tiles_ws(C.byref(d))
        stat = torch.zeros(max(tiles, 1) * 2 * d.Cout, device="cuda")
        nws = L.rd_gconv_workspace_floats(C.byref(d))
        ws = torch.empty(int(nws), device="cuda") if nws > 0 else None
        info = (C.c_int32 * 10)()
        L.rd_gconv_plan_info(C.byref(d), info)
        for use_add in (False, True):
            ref = None
            nbad = 0
            for it in range(reps):
                junk = torch.randn(1 << 20, device="cuda")          # disturb caches / timing between launches
                out.fill_(float("nan")) if d.out_stride == 1 and d.n_phases == 1 else out.zero_()
                check(L.rd_gconv_ws(C.byref(d), ptr(x), ptr(wp), ptr(out), ptr(add) if use_add else None, d.ldo, ptr(stat),
                                    ptr(ws) if ws is not None else None, current_stream()), name)
                torch.cuda.synchronize()
                cur = (out.clone(), stat.clone())
                if ref is None:
                    ref = cur
                elif not (torch.equal(ref[0].nan_to_num(), cur[0].nan_to_num()) and torch.equal(ref[1], cur[1])):
                    nbad += 1
            if nbad:
                bad += 1
                print("FLAKY %-40s add=%d: %d of %d launches differ; plan %s  N=%d %dx%d C %d->%d phases %d strides %d/%d"
                      % (name, use_add, nbad, reps, list(info), d.N, d.Hi, d.Wi, d.Cin, d.Cout, d.n_phases, d.in_stride, d.out_stride), flush=True)
print("%d distinct gconv descriptors, %d flaky" % (len(seen), bad))
