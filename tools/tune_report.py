"""What the plan tuner changes for one configuration: every tuned descriptor with the heuristic plan's time, the tuned plan's
time and both plans (MT, NT, WM, WN, CKP, TH, TW, ksplit, pipelined).  Planner rules are derived from this table.
    python tools/tune_report.py [arch] [batch] [height] [width]"""
import sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd import autotune
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import procedural_fill_

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet18_multistage_uncertainty_fixs"
b, h, w = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (8, 450, 800)))
args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
made = create_model(args, [h, w])
m, lw = made if isinstance(made, tuple) else (made, None)
procedural_fill_(m)
ts = HipTrainStep(m.cuda(), b, h, w, loss_weights=lw, autotune=True)
plans = [ts.mp.p1, ts.mp.p2] if getattr(ts, "mp", None) is not None else [ts.plan]
seen, rows = set(), []
for pl in plans:
    for name, (kind, d) in pl.meta.items():
        if kind not in ("gconv", "gconv_bnb"):
            continue
        key = bytes(d) + b"\x01"
        if key in seen or not autotune._TUNED.get(key):
            continue
        seen.add(key)
        best_us, heur_us, best_plan, heur_plan = autotune._TUNED[key]
        rows.append((heur_us - best_us, name, heur_us, best_us, heur_plan, best_plan, d))
rows.sort(key=lambda r: -r[0])
print("%-40s %9s %9s  %-34s %-34s" % ("descriptor", "heur us", "tuned us", "heuristic plan", "tuned plan"))
for gain, name, hu, bu, hp, bp, d in rows:
    print("%-40s %9.1f %9.1f  %-34s %-34s %s" % (name[:40], hu, bu, hp, bp, "<-- %.0f%%" % (100 * gain / hu) if gain > 0.05 * hu else ""))
print("sum heuristic %.1f us, tuned %.1f us over %d descriptors" % (sum(r[2] for r in rows), sum(r[3] for r in rows), len(rows)))
