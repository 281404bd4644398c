#!/usr/bin/env python3
"""Offline plan tuning (run on an MI355X): radar_depth_amd/tuned_plans.json.

For every distinct fp32 gconv descriptor (forward and input-gradient launches) of BASELINE.json's configurations the run-time tuner
lists and times the candidate plans (radar_depth_amd/autotune.py); a candidate that beat the heuristic is then re-timed against it
in alternating rounds and enters the table only if its MEDIAN stays >= 3 % faster.  The table is a pure function of the descriptor,
so -- unlike RD_AUTOTUNE=1 -- every process and every data-parallel rank pins the same plans.

    python tools/make_tuned_table.py [out.json]        # writes gpurun_out/tuned_plans.json by default; copy it into the package
"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RD_TUNED_TABLE"] = "0"          # tune against the plain heuristic

CONFIGS = [("resnet18_latefusion", 16, 450, 800), ("resnet18_multistage_uncertainty_fixs", 8, 450, 800),
           ("resnet18_multistage_uncertainty_fixs", 8, 900, 1600), ("resnet18_latefusion", 16, 900, 1600)]


def _descriptors_here(arch, b, h, w):
    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.model.multistage_model import ResNet_latefusion2
    import torch
    out = []
    if arch == "resnet18_latefusion":
        nets = [ResNet_latefusion(18, "upproj", [h, w], 4, False)]
    else:
        nets = [ResNet_latefusion2(18, "upproj", [h, w], 4, False), ResNet_latefusion2(18, "upproj", [h, w], 5, False)]
    for k, m in enumerate(nets):
        dp = [torch.empty(b, h, w), torch.empty(b, h, w)] if k == 1 else None
        plan = LateFusionPlan(m, b, h, w, train=True, dry_run=True, depth_planes=dp)
        for name, (kind, d) in plan.meta.items():
            if kind in ("gconv", "gconv_bnb"):
                out.append((name, d))
    return out


def descriptors(arch, b, h, w):
    """Enumerated in a CHILD process: building a plan queries the library's planner, which caches the heuristic plan of every
    descriptor as "in use" -- the tuner then (rightly) refuses to replace it in that process."""
    import subprocess
    from radar_depth_amd._lib import RdConvDesc
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--list", arch, str(b), str(h), str(w)], capture_output=True, text=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    out = []
    for ln in r.stdout.splitlines():
        if ln.startswith("DESC\t"):
            _, name, hx = ln.split("\t")
            out.append((name, RdConvDesc.from_buffer_copy(bytes.fromhex(hx))))
    assert out, r.stderr[-2000:]
    return out


def main():
    import torch
    from radar_depth_amd import autotune as at
    from radar_depth_amd._lib import lib
    L = lib()
    dev = torch.device("cuda", 0)
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "tuned_plans.json")
    plans, seen = {}, set()
    tot_h = tot_t = 0.0
    for arch, b, h, w in CONFIGS:
        for name, d in descriptors(arch, b, h, w):
            key = at.desc_key(d)
            if key in seen:
                continue
            seen.add(key)
            res = at.tune_gconv(L, d, dev)
            L.rd_gconv_tune_pin(C.byref(d), 1, None)
            if not res:
                continue
            best_us, heur_us, best_plan, heur_plan = res
            if tuple(best_plan) == tuple(heur_plan or ()) or best_us > 0.97 * heur_us:
                continue
            t_heur, t_best = at.time_plans(L, d, dev, [None, list(best_plan)])
            L.rd_gconv_tune_pin(C.byref(d), 1, None)
            keep = t_best <= 0.97 * t_heur
            print("%-9s b=%d %dx%d %-40s heuristic %7.1f us %-28s tuned %7.1f us %-28s %s" % (
                arch[9:18], b, h, w, name, t_heur, tuple(heur_plan), t_best, tuple(best_plan), "KEEP" if keep else "drop"), flush=True)
            if keep:
                plans[key] = {"plan": [int(v) for v in best_plan], "us": round(t_best, 1), "heuristic_us": round(t_heur, 1),
                              "layer": "%s b=%d %dx%d %s" % (arch, b, h, w, name)}
                tot_h += t_heur
                tot_t += t_best
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    with open(out_path, "w") as f:
        json.dump({"note": "offline-tuned gconv plans (tools/make_tuned_table.py on an MI355X): plan = MT, NT, WM, WN, CKP, TH, TW, ksplit, "
                           "pipelined; kept only where the median of alternating timings beats the heuristic by >= 3 %",
                   "device": torch.cuda.get_device_name(0), "plans": plans}, f, indent=1, sort_keys=True)
    print("%d descriptors seen, %d plans kept: sum heuristic %.1f us -> tuned %.1f us; written to %s" % (len(seen), len(plans), tot_h, tot_t, out_path))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--list":
        for name, d in _descriptors_here(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])):
            print("DESC\t%s\t%s" % (name, bytes(d).hex()))
    else:
        main()
