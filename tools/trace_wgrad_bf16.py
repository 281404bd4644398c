#!/usr/bin/env python3
"""Phase timeline of rd_wgrad_bf16 workgroups (RD_WGRAD_BF16_TRACE=1): median clocks between stamps
(MFMA wave: start | per tile: buffer ready, MFMAs done | slab written; staging wave (RD_WGRAD_BF16_TRACE=2): per tile: buffer
handed over, loads issued, tile written to LDS).
    python tools/trace_wgrad_bf16.py <layer index of tools/bench_ops.py CONVS>"""
import ctypes as C, os, sys
os.environ.setdefault("RD_WGRAD_BF16_TRACE", "1")      # 1: an MFMA wave, 2: a staging wave
sys.path.insert(0, ".")
import numpy as np, torch
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import lib
idx = int(sys.argv[1]); sys.argv = sys.argv[:1]
from tools.bench_ops import CONVS
B = 16
name, cnt, ci, co, k, s, p, h, w = CONVS[idx]
d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
x = torch.randn(B, h, w, ci, device="cuda")
y = torch.randn(B, d.Ho, d.Wo, co, device="cuda")
n = int(lib().rd_wgrad_bf16_workspace_floats(C.byref(d)))
slabs = torch.empty(n, device="cuda")
for _ in range(100):
    lib().rd_wgrad_bf16(C.byref(d), ops.ptr(x), ops.ptr(y), ops.ptr(slabs), ops.current_stream())
torch.cuda.synchronize()
info = (C.c_int32 * 8)()
lib().rd_wgrad_bf16_plan_info(C.byref(d), info)
nwg = info[2] * info[3] * info[6]
buf = np.zeros(nwg * 32, dtype=np.uint64)
assert lib().rd_wgrad_bf16_trace_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
t = buf.reshape(nwg, 32)
ns = int(np.median(t[:, 0]))
st = t[t[:, 0] == ns][:, 1:1 + ns].astype(np.int64)
dd = np.diff(st, axis=1)
print(name, "workgroups", nwg, "stamps", ns)
print("median clocks between stamps:", " ".join("%d" % v for v in np.median(dd, axis=0)))
print("median lifetime %d clk" % np.median(st[:, -1] - st[:, 0]))
