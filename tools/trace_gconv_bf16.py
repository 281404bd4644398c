#!/usr/bin/env python3
"""Phase timeline of rd_gconv_bf16 workgroups (RD_GCONV_BF16_TRACE=1): median clocks between the stamps
(start | prologue | per chunk: staged+barrier, MFMA walk | epilogue).  python tools/trace_gconv_bf16.py <layer index of bench_ops>"""
import ctypes as C, os, sys
os.environ["RD_GCONV_BF16_TRACE"] = "1"
sys.path.insert(0, ".")
import numpy as np, torch
from radar_depth_amd import convdesc as cd, ops
from radar_depth_amd._lib import lib
IO16 = "--io16" in sys.argv          # bf16-storage form (rd_gconv_bf16_t with RD_DTYPE_BF16)
sys.argv = [a for a in sys.argv if a != "--io16"]
sys.argv, idx = sys.argv[:1], int(sys.argv[1])
from tools.bench_ops import CONVS
from tools.bench_ops_bf16 import plan
B = 16
name, cnt, ci, co, k, s, p, h, w = CONVS[idx]
d = cd.conv_fwd(B, h, w, ci, co, k, s, p)
from radar_depth_amd._lib import current_stream, ptr
adt = torch.bfloat16 if IO16 else torch.float32
x = torch.randn(B, h, w, ci, device="cuda").to(adt)
wp = ops.pack_weights_bf16(torch.randn(co, ci, k, k, device="cuda"))
y = torch.empty(B, d.Ho, d.Wo, co, device="cuda", dtype=adt)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(200):
    if it == 100:
        ev0.record()
    assert lib().rd_gconv_bf16_t(1 if IO16 else 0, C.byref(d), ptr(x), ptr(wp), ptr(y), None, 0, 0, None, 0, None, current_stream()) == 0
ev1.record()
torch.cuda.synchronize()
print("io16" if IO16 else "fp32 io", "avg launch %.1f us" % (ev0.elapsed_time(ev1) * 10.0))
torch.cuda.synchronize()
info = (C.c_int32 * 8)()
lib().rd_gconv_bf16_plan_info(C.byref(d), info)
nwg = info[7]
buf = np.zeros(nwg * 32, dtype=np.uint64)
assert lib().rd_gconv_bf16_trace_read(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), nwg) == 0
t = buf.reshape(nwg, 32)
n = int(t[0, 0])
st = t[:, 1:1 + n].astype(np.int64)
dd = np.diff(st, axis=1)
print(name, plan(d), "stamps", n)
print("median clocks between stamps:", " ".join("%d" % v for v in np.median(dd, axis=0)))
life = st[:, -1] - st[:, 0]
print("workgroup lifetime clk: p10 %d  p50 %d  p90 %d  max %d  mean %d   (%d workgroups; x rounds / 2.38 GHz ~ kernel time)"
      % (np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max(), life.mean(), nwg))
print("p90 clocks between stamps:   ", " ".join("%d" % v for v in np.percentile(dd, 90, axis=0)))
rt = t[:, 31].astype(np.int64)          # workgroup lifetime on the 100 MHz constant clock
print("shader clock while this kernel runs: %.2f GHz (cycle counter / 100 MHz real-time counter over the workgroup lifetimes)"
      % (life.sum() / (rt.sum() * 10.0) ))
