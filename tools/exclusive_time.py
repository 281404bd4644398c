#!/usr/bin/env python3
"""Where the step is serial: from a rocprofv3 kernel trace (rocpd sqlite), per kernel name the time during which it is the ONLY kernel running
(nothing overlaps it: what a shorter kernel would give back one for one), the time it shares the device, and the idle time between kernels.
Steady state only (from the first sgd_kernel on).   python tools/exclusive_time.py x_results.db [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
kcols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else "name")
rows = db.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id" % (name_col, kd, ks)).fetchall()
t0 = min(st for name, st, en in rows if "sgd_kernel" in name)
rows = [(re.sub(r"\(.*$", "", n).replace("void rd::", "").replace("rd::", ""), st, en) for n, st, en in rows if st >= t0]
ev = []
for i, (n, st, en) in enumerate(rows):
    ev.append((st, 1, i))
    ev.append((en, 0, i))
ev.sort()
active = set()
excl, shared = {}, {}
idle = 0
prev = ev[0][0]
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        if not active:
            idle += dt
        elif len(active) == 1:
            n = rows[next(iter(active))][0]
            excl[n] = excl.get(n, 0) + dt
        else:
            for j in active:
                n = rows[j][0]
                shared[n] = shared.get(n, 0) + dt / len(active)
    prev = t
    if kind:
        active.add(i)
    else:
        active.discard(i)
span = ev[-1][0] - ev[0][0]
print("# %s: %d dispatches, span %.3f ms, idle %.3f ms (%.1f %%), exclusive %.3f ms, shared %.3f ms; per step (/%d)" % (
    sys.argv[1].split("/")[-1], len(rows), span / 1e6, idle / 1e6, 100.0 * idle / span, sum(excl.values()) / 1e6, sum(shared.values()) / 1e6, steps))
print("%-62s %12s %12s" % ("kernel", "exclusive us", "shared us"))
for n in sorted(set(excl) | set(shared), key=lambda k: -excl.get(k, 0))[:45]:
    print("%-62s %12.1f %12.1f" % (n[:62], excl.get(n, 0) / 1e3 / steps, shared.get(n, 0) / 1e3 / steps))
