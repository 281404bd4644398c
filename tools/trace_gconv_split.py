#!/usr/bin/env python3
"""Where a gconv_split workgroup spends its clocks: per tap group, the wait at the barrier and the work between barriers, for an MFMA
wave (RD_GCONV_SPLIT_TRACE=1) or a staging wave (=2).  Run on the GPU box:  RD_GCONV_SPLIT_TRACE=1 python tools/trace_gconv_split.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "tools")
from radar_depth_amd import convdesc as cd, ops  # noqa: E402
from radar_depth_amd._lib import lib  # noqa: E402
from bench_split import plan  # noqa: E402

B = 16
PRE = "--pre" in sys.argv          # rd_gconv_split_pre (activation split by its producer) instead of rd_gconv_split
role = os.environ.get("RD_GCONV_SPLIT_TRACE", "0")
for name, ci, co, k, h, w in [("layer1", 64, 64, 3, 113, 200), ("layer2", 128, 128, 3, 57, 100), ("layer3", 256, 256, 3, 29, 50), ("layer4", 512, 512, 3, 15, 25)]:
    d = cd.conv_fwd(B, h, w, ci, co, k, 1, 1)
    x = torch.randn(B, h, w, ci, device="cuda"); wt = torch.randn(co, ci, k, k, device="cuda")
    y = torch.empty(B, h, w, co, device="cuda")
    ws = ops.pack_weights_split(wt)
    xp = ops.split_pieces(x) if PRE else None
    for _ in range(20):
        if PRE:
            ops.gconv_split_pre(d, xp, ws, y)
        else:
            ops.gconv_split(d, x, ws, y)
    torch.cuda.synchronize()
    v = (C.c_int32 * 8)()
    (lib().rd_gconv_split_pre_plan_info if PRE else lib().rd_gconv_split_plan_info)(C.byref(d), v)
    nwg = min(int(v[6]), 65536)
    buf = np.zeros((nwg, 64), dtype=np.uint64)
    assert lib().rd_gconv_split_trace_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
    t = buf.astype(np.int64)
    ng = int(t[0, 63]) if role == "1" else 30
    n = min(ng, 30)
    wait = (t[:, 1:2 * n:2] - t[:, 0:2 * n:2])                     # barrier wait per group
    work = (t[:, 2:2 * n:2] - t[:, 1:2 * n - 1:2])                 # barrier exit -> next barrier entry
    print("%-7s %s%s | role %s | groups/wg %d" % (name, "PRE " if PRE else "", "%dx%d %dx%d lds %dK wg %d" % (v[0], v[1], v[2], v[3], v[5] // 1024, v[6]), role, ng))
    print("   prologue (entry -> first barrier entry) median %d clk" % np.median(t[:, 0] - t[:, 62]))
    print("   barrier wait per group: median %d, mean %d, p90 %d ; first group %d" % (np.median(wait[:, 1:]), wait[:, 1:].mean(), np.percentile(wait[:, 1:], 90), np.median(wait[:, 0])))
    print("   work between barriers : median %d, mean %d, p90 %d clk   (MFMA floor: %d)" % (np.median(work), work.mean(), np.percentile(work, 90), 3 * v[0] * v[1] * 6 * 32))
    per = np.median(work, axis=0)
    print("   per group (median over workgroups): " + " ".join("%d" % p for p in per[:12]))
    if PRE and role == "1":
        # residency: wall-clock (100 MHz) entry / exit of every workgroup, grouped by the CU it ran on (XCC_ID, HW_ID: se, sh, cu)
        t0, t1, hw = t[:, 59], t[:, 58], buf[:, 57]
        xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
        hid = hw & np.uint64(0xffffffff)
        cu = (xcc.astype(np.int64) << 16) | (((hid >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64))       # cu_id[11:8] sh[12] se[15:13]
        base = t0.min()
        span = (t1.max() - base) / 100.0
        res = []
        for c in np.unique(cu):
            sel = cu == c
            ev = sorted([(a_, 1) for a_ in t0[sel]] + [(b_, -1) for b_ in t1[sel]])
            cur = peak = 0; busy2 = 0; last = None
            for tm, dlt in ev:
                if last is not None and cur >= 2:
                    busy2 += tm - last
                cur += dlt; peak = max(peak, cur); last = tm
            res.append((int(sel.sum()), peak, busy2 / 100.0))
        r = np.array(res)
        print("   residency: %d CUs used; workgroups per CU min %d max %d; peak co-resident per CU: min %d max %d; us with >= 2 resident (median over CUs) %.1f of a %.1f us kernel"
              % (len(r), r[:, 0].min(), r[:, 0].max(), r[:, 1].min(), r[:, 1].max(), np.median(r[:, 2]), span))
    if role == "1":
        print("   MFMA loop total median %d clk, epilogue median %d clk" % (np.median(t[:, 60] - t[:, 0]), np.median(t[:, 61] - t[:, 60])))
