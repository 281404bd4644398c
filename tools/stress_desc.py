"""Race hunt at the level of single launches: every convolution descriptor of a model's plans (forward, dgrad, weight gradient; fp32,
bf16 operands or bf16 storage) is launched REPS times on the same inputs; outputs, BatchNorm partial sums and weight-gradient
slabs must be bit-identical every time.
    python tools/stress_desc.py [arch] [batch] [height] [width] [reps] [fp32|bf16|bf16s]"""
import ctypes as C, sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd._lib import check, current_stream, lib, ptr
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import procedural_fill_

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18_multistage_uncertainty_fixs"
b, h, w = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (2, 450, 800)))
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 25
MODE = sys.argv[6] if len(sys.argv) > 6 else "fp32"
OPERANDS, STORAGE = ("bf16" if MODE != "fp32" else "fp32"), ("bf16" if MODE == "bf16s" else "fp32")
IO16 = 1 if STORAGE == "bf16" else 0
adt = torch.bfloat16 if IO16 else torch.float32
L = lib()
L.rd_gconv_workspace_floats.restype = C.c_int64
args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
made = create_model(args, [h, w])
m, lw = made if isinstance(made, tuple) else (made, None)
procedural_fill_(m)
ts = HipTrainStep(m.cuda(), b, h, w, loss_weights=lw, operands=OPERANDS, storage=STORAGE)
L.rd_wgrad_workspace_floats.restype = C.c_int64
L.rd_wgrad_bf16_workspace_floats.restype = C.c_int64
plans = [ts.mp.p1, ts.mp.p2] if getattr(ts, "mp", None) is not None else [ts.plan]
seen, bad = set(), 0
for pl in plans:
    for name, (kind, d) in pl.meta.items():
        key = bytes(d) + kind.encode()
        if key in seen:
            continue
        seen.add(key)
        S = max(d.phase[i].widx[t] for i in range(d.n_phases) for t in range(d.phase[i].n_taps)) + 1
        g = torch.Generator(device="cuda").manual_seed(1)
        if kind in ("wgrad", "wgrad_bf16"):
            dt = adt if kind == "wgrad_bf16" else torch.float32
            x = torch.randn(d.N * d.Hi * d.Wi * d.ldi, device="cuda", generator=g).to(dt)
            dy = torch.randn(d.N * d.Ho * d.Wo * d.ldo, device="cuda", generator=g).to(dt)
            nws = int((L.rd_wgrad_bf16_workspace_floats if kind == "wgrad_bf16" else L.rd_wgrad_workspace_floats)(C.byref(d)))
            slabs = torch.zeros(nws, device="cuda")
            ref, nbad = None, 0
            for it in range(reps):
                junk = torch.randn(1 << 20, device="cuda")
                slabs.fill_(float("nan"))
                if kind == "wgrad_bf16":
                    check(L.rd_wgrad_bf16_t(IO16, C.byref(d), ptr(x), ptr(dy), ptr(slabs), current_stream()), name)
                else:
                    check(L.rd_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(slabs), current_stream()), name)
                torch.cuda.synchronize()
                cur = slabs.nan_to_num(1e30).clone()
                if ref is None:
                    ref = cur
                elif not torch.equal(ref, cur):
                    nbad += 1
            if nbad:
                bad += 1
                print("FLAKY %-40s %s: %d of %d launches differ  N=%d %dx%d C %d->%d phases %d strides %d/%d"
                      % (name, kind, nbad, reps, d.N, d.Hi, d.Wi, d.Cin, d.Cout, d.n_phases, d.in_stride, d.out_stride), flush=True)
            continue
        bf = kind == "gconv_bf16"
        dt = adt if bf else torch.float32
        x = torch.randn(d.N * d.Hi * d.Wi * d.ldi, device="cuda", generator=g).to(dt)
        wp = torch.randn(S * d.Cin * d.Cout, device="cuda", generator=g)
        if bf:
            wp = wp.to(torch.bfloat16)
        out = torch.zeros(d.N * d.Ho * d.Wo * d.ldo, device="cuda", dtype=dt)
        add = torch.randn(d.N * d.Ho * d.Wo * d.ldo, device="cuda", generator=g).to(dt)
        tiles = (L.rd_gconv_bf16_stat_tiles if bf else L.rd_gconv_stat_tiles_ws)(C.byref(d))      # (fp32 tensors: rd_gconv_bf16_t with bf16 tensors sizes with rd_gconv_bf16_stat_tiles_t)
        stat = torch.zeros(max(tiles, 1) * 2 * d.Cout, device="cuda")
        nws = 0 if bf else L.rd_gconv_workspace_floats(C.byref(d))
        ws = torch.empty(int(nws), device="cuda") if nws > 0 else None
        info = (C.c_int32 * 10)()
        (L.rd_gconv_bf16_plan_info if bf else L.rd_gconv_plan_info)(C.byref(d), info)
        for use_add in (False, True):
            ref = None
            nbad = 0
            for it in range(reps):
                junk = torch.randn(1 << 20, device="cuda")          # disturb caches / timing between launches
                out.fill_(float("nan")) if d.out_stride == 1 and d.n_phases == 1 else out.zero_()
                if bf:
                    check(L.rd_gconv_bf16_t(IO16, C.byref(d), ptr(x), ptr(wp), ptr(out), None, 0, 0, ptr(add) if use_add else None, d.ldo,
                                            ptr(stat), current_stream()), name)
                else:
                    check(L.rd_gconv_ws(C.byref(d), ptr(x), ptr(wp), ptr(out), ptr(add) if use_add else None, d.ldo, ptr(stat),
                                        ptr(ws) if ws is not None else None, current_stream()), name)
                torch.cuda.synchronize()
                cur = (out.float().nan_to_num(1e30).clone(), stat.clone())
                if ref is None:
                    ref = cur
                elif not (torch.equal(ref[0], cur[0]) and torch.equal(ref[1], cur[1])):
                    nbad += 1
            if nbad:
                bad += 1
                print("FLAKY %-40s %s add=%d: %d of %d launches differ; plan %s  N=%d %dx%d C %d->%d phases %d strides %d/%d"
                      % (name, kind, use_add, nbad, reps, list(info)[:8], d.N, d.Hi, d.Wi, d.Cin, d.Cout, d.n_phases, d.in_stride, d.out_stride), flush=True)
print("%s %s b=%d %dx%d: %d distinct descriptors, %d flaky" % (MODE, arch, b, h, w, len(seen), bad))
