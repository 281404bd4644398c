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
tot = sum(a[1] for a in agg.values())
print("# rocprofv3 --kernel-trace summary of %s (%d dispatches, %.3f ms of kernel time)" % (sys.argv[1], len(rows), tot / 1e6))
print("%-72s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "share"))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %8d %12.3f %10.2f %6.1f%%" % (name[:72], n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
