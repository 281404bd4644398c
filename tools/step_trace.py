#!/usr/bin/env python3
"""One training step of a rocprofv3 (rocpd sqlite) kernel trace as a per-queue timeline: which kernels of which stream are
running when, and what the tail of the step looks like (the depth-encoder chain of the last backward segment).
    python tools/step_trace.py tr_results.db [step_index_from_the_end=2] > step_trace.txt
Columns: start offset (us, from the end of the previous step's sgd_kernel), duration (us), queue, number of OTHER dispatches running at the
kernel's start, kernel name.  The summary lists, per queue, the busy time and the time it is the ONLY busy queue."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
kcols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
name_col = "display_name" if "display_name" in kcols else ("kernel_name" if "kernel_name" in kcols else "name")
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select s.%s, d.start, d.end, %s from %s d join %s s on d.kernel_id = s.id order by d.start"
                  % (name_col, "d." + qcol if qcol else "0", kd, ks)).fetchall()


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void rd::", "").replace("rd::", "")
    return n[:64]


sgd = [(st, en) for n, st, en, q in rows if "sgd_kernel" in n]
assert len(sgd) > back + 1, "trace holds too few steps"
t0, t1 = sgd[-back - 1][1], sgd[-back][1]
step = [(short(n), st, en, q) for n, st, en, q in rows if st >= t0 and en <= t1]
queues = sorted(set(q for _, _, _, q in step))
qname = {q: "q%d" % i for i, q in enumerate(queues)}
print("# step of %.3f ms, %d dispatches, queues %s (columns of %s: %s)" % ((t1 - t0) / 1e6, len(step), [qname[q] for q in queues], kd, qcol))
ev = []
for n, st, en, q in step:
    ev.append((st, 1, q)); ev.append((en, -1, q))
ev.sort()
# per-queue busy time, time as the only busy queue, idle time
active = {q: 0 for q in queues}
last = t0
busy = {q: 0 for q in queues}
only = {q: 0 for q in queues}
idle = 0
for t, d, q in ev:
    on = [k for k in queues if active[k] > 0]
    dt = t - last
    if not on:
        idle += dt
    for k in on:
        busy[k] += dt
    if len(on) == 1:
        only[on[0]] += dt
    active[q] += d
    last = t
print("# idle %.1f us; per queue busy / only-busy us: %s" % (idle / 1e3, "  ".join("%s %.0f / %.0f" % (qname[q], busy[q] / 1e3, only[q] / 1e3) for q in queues)))
for n, st, en, q in step:
    others = sum(1 for _, a, b, _ in step if a <= st < b) - 1
    print("%9.1f %8.1f %s %d %s" % ((st - t0) / 1e3, (en - st) / 1e3, qname[q], others, n))
