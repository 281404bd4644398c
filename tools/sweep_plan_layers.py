#!/usr/bin/env python3
"""Planner audit: time every distinct gconv descriptor of the b=16 450x800 training plan (forward and input-gradient launches) under
the planner's own choice and under forced alternatives; every (tile, CKP, split, pipe) point runs in its own interpreter because the
planner reads its overrides once per process.
    python tools/sweep_plan_layers.py            # parent: prints, per descriptor, auto vs the best forced point
    python tools/sweep_plan_layers.py --child    # child: one line per descriptor (used by the parent)"""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def descriptors():
    import torch
    from radar_depth_amd.engine import LateFusionPlan
    from radar_depth_amd.model.models import ResNet_latefusion
    m = ResNet_latefusion(18, "upproj", [450, 800], 4, False)
    plan = LateFusionPlan(m, 16, 450, 800, train=True, dry_run=True)
    seen, out = set(), []
    for name, (kind, d) in plan.meta.items():
        if kind not in ("gconv", "gconv_bnb"):
            continue
        key = bytes(d)
        if key in seen:
            continue
        seen.add(key)
        out.append((name, d))
    return out


def child():
    import torch
    from radar_depth_amd import ops
    from radar_depth_amd._lib import lib
    L = lib()
    for name, d in descriptors():
        info = (C.c_int32 * 10)()
        if L.rd_gconv_plan_info(C.byref(d), info) != 0:
            print("%s\t-\t-" % name)
            continue
        S = max(max(d.phase[i].widx[t] for t in range(d.phase[i].n_taps)) for i in range(d.n_phases)) + 1
        x = torch.randn(d.N, d.Hi, d.Wi, d.ldi, device="cuda")
        wp = torch.randn(S, d.Cin, d.Cout, device="cuda")
        y = torch.empty(d.N, d.Ho, d.Wo, d.ldo, device="cuda")
        try:
            w0 = time.perf_counter()
            while time.perf_counter() - w0 < 0.04:
                for _ in range(10):
                    ops.gconv(d, x, wp, y)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gconv(d, x, wp, y)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100.0
        except Exception as ex:   # noqa: BLE001
            us = float("inf")
        print("%s\t%.1f\t(%d,%d) ckw%d ckp%d %dx%d ks%d pipe%d grp%d" % (name, us, info[0], info[1], info[4] % 100, info[5], info[6], info[7],
                                                                        (info[4] // 100) % 100, (info[4] // 10000) % 100, info[4] // 1000000))


def main():
    if "--child" in sys.argv:
        return child()
    points = [{}]
    for force in ("0", "1", "2", "3", "4"):
        for ckp in ("16", "32"):
            for nosplit in (False, True):
                for nopipe in (False, True):
                    e = {"RD_GCONV_FORCE": force, "RD_GCONV_CKP": ckp}
                    if nosplit:
                        e["RD_GCONV_NOSPLIT"] = "1"
                    if nopipe:
                        e["RD_GCONV_NOPIPE"] = "1"
                    points.append(e)
    results = {}
    for e in points:
        r = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **e), capture_output=True, text=True)
        for ln in r.stdout.splitlines():
            parts = ln.split("\t")
            if len(parts) == 3 and parts[1] != "-":
                results.setdefault(parts[0], []).append((float(parts[1]), parts[2], "auto" if not e else " ".join("%s=%s" % kv for kv in e.items())))
    tot_auto = tot_best = 0.0
    for name, rows in results.items():
        auto = [r for r in rows if r[2] == "auto"][0]
        best = min(rows)
        tot_auto += auto[0]
        tot_best += best[0]
        flag = "  <-- %.0f%%" % (100 * (auto[0] / best[0] - 1)) if auto[0] > 1.05 * best[0] else ""
        print("%-44s auto %7.1f us %-40s best %7.1f us %-40s%s" % (name, auto[0], auto[1], best[0], best[1], flag))
    print("sum over distinct descriptors: auto %.1f us, best %.1f us" % (tot_auto, tot_best))


if __name__ == "__main__":
    main()
