"""Per-layer audit of the fp32 weight-gradient planner's knobs (the wgrad counterpart of tools/sweep_plan_layers.py): every distinct
weight-gradient descriptor of the bench step is timed (kernel + slab reduction) under each setting of the planner's diagnostic
environment variables -- they are read once per process, so every setting runs in a child process.
    python tools/sweep_wgrad_knobs.py"""
import ctypes as C, json, os, subprocess, sys, types
SETTINGS = [("default", {}), ("no strip", {"RD_WGRAD_NOSTRIP": "1"}), ("1 wg/cu", {"RD_WGRAD_WG_PER_CU_X2": "2"}),
            ("1.5 wg/cu", {"RD_WGRAD_WG_PER_CU_X2": "3"}), ("3 wg/cu", {"RD_WGRAD_WG_PER_CU_X2": "6"}),
            ("lds 40K", {"RD_WGRAD_LDS_KB": "40"}), ("lds 120K", {"RD_WGRAD_LDS_KB": "120"}), ("generic 16ch", {"RD_WGRAD_NOW16": "1"})]


def child():
    import torch
    sys.path.insert(0, ".")
    from radar_depth_amd._lib import check, current_stream, lib, ptr
    from radar_depth_amd.main import HipTrainStep, create_model
    from radar_depth_amd.synthetic import procedural_fill_
    from tools.bench_ops import timeit
    L = lib()
    L.rd_wgrad_workspace_floats.restype = C.c_int64
    args = types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    m = create_model(args, [450, 800])
    procedural_fill_(m)
    ts = HipTrainStep(m.cuda(), 16, 450, 800)
    out, seen = {}, set()
    for name, (kind, d) in ts.plan.meta.items():
        if kind != "wgrad" or bytes(d) in seen:
            continue
        seen.add(bytes(d))
        S = max(d.phase[i].widx[t] for i in range(d.n_phases) for t in range(d.phase[i].n_taps)) + 1
        x = torch.randn(d.N * d.Hi * d.Wi * d.ldi, device="cuda")
        dy = torch.randn(d.N * d.Ho * d.Wo * d.ldo, device="cuda")
        ws = torch.empty(int(L.rd_wgrad_workspace_floats(C.byref(d))), device="cuda")
        kk = int(round(S ** 0.5))
        grad = torch.empty(d.Cout * d.Cin * S, device="cuda")

        def run():
            check(L.rd_wgrad(C.byref(d), ptr(x), ptr(dy), ptr(ws), current_stream()), name)
            check(L.rd_wgrad_reduce(C.byref(d), ptr(ws), ptr(grad), d.Cout, d.Cin, kk, S // kk, 0, 0, current_stream()), name)
        out[name] = timeit(run) * 1e6
    print("RESULT " + json.dumps(out))


if "--child" in sys.argv:
    child()
    sys.exit(0)
table = {}
for label, env in SETTINGS:
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, __file__, "--child"], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        print(label, "FAILED", r.stderr[-400:])
        continue
    table[label] = json.loads(line[0][7:])
names = list(table["default"])
print("%-44s" % "layer (us: kernel + reduce)" + "".join("%12s" % l for l, _ in SETTINGS if l in table))
tot = {l: 0.0 for l in table}
best_tot = 0.0
for n in names:
    row = [table[l].get(n, float("nan")) for l, _ in SETTINGS if l in table]
    b = min(row)
    best_tot += b
    for l in table:
        tot[l] += table[l].get(n, 0.0)
    print("%-44s" % n[:44] + "".join("%12.1f" % v for v in row) + ("   <-- %s %.0f%%" % ([l for l, _ in SETTINGS if l in table][row.index(b)], 100 * (row[0] - b) / row[0]) if b < 0.95 * row[0] else ""))
print("%-44s" % "sum over distinct descriptors" + "".join("%12.1f" % tot[l] for l, _ in SETTINGS if l in table) + "   best-per-layer %.1f" % best_tot)
