"""Effective shader clock while the training step runs continuously: RD_GCONV_TRACE=1 stamps every gconv workgroup with the cycle
counter and the 100 MHz real-time counter; after N steps the buffer holds the last gconv launch of the step."""
import ctypes as C, os, sys, types
import numpy as np, torch
os.environ["RD_GCONV_TRACE"] = "1"
sys.path.insert(0, ".")
from radar_depth_amd._lib import lib
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import make_batch
args = types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
m = create_model(args, [450, 800]).cuda()
ts = HipTrainStep(m, 16, 450, 800)
x, t = make_batch(16, 450, 800, 1); x, t = x.cuda(), t.cuda()
for _ in range(40): ts.step(x, t)
torch.cuda.synchronize()
L = lib()
nwg = 256
buf = np.zeros((nwg, 64), dtype=np.uint64)
L.rd_gconv_trace_read.argtypes = [C.c_void_p, C.c_int]
assert L.rd_gconv_trace_read(buf.ctypes.data, nwg) == 0
n = int(buf[0, 0])
cyc = (buf[:, n] - buf[:, 1]).astype(np.float64)
rt = buf[:, 62].astype(np.float64) / 100e6
ok = (rt > 0) & (cyc > 0)
print("last gconv launch of the step, %d workgroups: effective shader clock %.0f MHz (min %.0f max %.0f)" % (
    ok.sum(), np.mean(cyc[ok] / rt[ok]) / 1e6, np.min(cyc[ok] / rt[ok]) / 1e6, np.max(cyc[ok] / rt[ok]) / 1e6))
