#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
kcols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else "name")
rows = db.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)).fetchall()
steady = ""
if "--steady" in sys.argv:     # training steps only: drop everything before the first optimizer update (plan building, weight
    t0 = min((st for name, st, en in rows if "sgd_kernel" in name), default=None)     # packing, the plan tuner's timing launches)
    if t0 is not None:
        n_all = len(rows)
        rows = [r for r in rows if r[1] >= t0]
        steady = "; steady state only: %d of %d dispatches, from the first sgd_kernel on" % (len(rows), n_all)
agg = {}
for name, st, en in rows:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void rd::", "").replace("rd::", "")
    a = agg.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += en - st
if "--top" in sys.argv:      # longest single dispatches of the kernels whose name contains the given substring
    sub = sys.argv[sys.argv.index("--top") + 1]
    sel = sorted(((en - st) / 1e3, re.sub(r"\(.*$", "", name)) for name, st, en in rows if sub in name)
    print("# %d dispatches matching %r; durations (us), descending" % (len(sel), sub))
    print(" ".join("%.1f" % d for d, _ in sel[::-1][:80]))
    sys.exit(0)
if "--gaps" in sys.argv:       # which kernels end before / start after the idle gaps (steady-state part of the trace)
    iv = sorted((st, en, re.sub(r"\(.*$", "", name).replace("void rd::", "").replace("rd::", "")) for name, st, en in rows)
    t_lo = iv[0][0] + 0.4 * (iv[-1][1] - iv[0][0])
    iv = [x for x in iv if x[0] >= t_lo]
    pairs = {}
    cur_end, cur_name = iv[0][1], iv[0][2]
    for a, b, nm in iv[1:]:
        if a > cur_end + 1500:
            k = (cur_name[:40], nm[:40])
            v = pairs.setdefault(k, [0, 0])
            v[0] += 1; v[1] += a - cur_end
        if b > cur_end:
            cur_end, cur_name = b, nm
    print("# idle gaps > 1.5 us: (kernel ending last before the gap -> kernel starting after it): count, total us")
    for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:25]:
        print("%-42s -> %-42s %5d %9.1f" % (k[0], k[1], v[0], v[1] / 1e3))
    sys.exit(0)
if "--timeline" in sys.argv:   # how full is the device: union of kernel intervals vs wall span, over the last 60 % of the trace
    iv = sorted((st, en) for _, st, en in rows)
    t_lo = iv[0][0] if steady else iv[0][0] + 0.4 * (iv[-1][1] - iv[0][0])
    iv = [(a, b) for a, b in iv if a >= t_lo]
    span = iv[-1][1] - iv[0][0]
    busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
    gaps = []
    for a, b in iv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            gaps.append(a - cur_b)
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    ssum = sum(b - a for a, b in iv)
    print("# steady-state window: %.2f ms wall, %.2f ms with at least one kernel running (%.1f %%), %.2f ms summed kernel time "
          "(average concurrency %.2f), %d idle gaps totalling %.3f ms (median %.2f us)" % (
              span / 1e6, busy / 1e6, 100.0 * busy / span, ssum / 1e6, ssum / busy, len(gaps), sum(gaps) / 1e6,
              sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0))
    sys.exit(0)
tot = sum(a[1] for a in agg.values())
print("# rocprofv3 --kernel-trace summary of %s (%d dispatches, %.3f ms of kernel time%s) -- collected at git %s"
      % (sys.argv[1], len(rows), tot / 1e6, steady, __import__("os").environ.get("RD_HEAD", "unknown")))
print("%-72s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %8d %12.3f %10.2f %6.1f%%" % (name[:72], n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
