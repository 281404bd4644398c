"""Host issue time of one training step against its wall time: the host must stay ahead of the device.
    python tools/host_time.py B H W {fp32|bf16|bf16s}        (tiny sizes -> pure host cost per step: the GPU never back-pressures)
Reports the op-table path (ONE rd_optable_run call per step, the default) and the Python loop over the same ops (RD_PY_LOOP=1),
wall and CPU time of the issuing thread per step."""
import os, sys, time, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import make_batch
args = types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 450, 800)))
MODE = sys.argv[4] if len(sys.argv) > 4 else "fp32"                                       # fp32 | bf16 (operands) | bf16s (storage)
m = create_model(args, [H, W]).cuda()
ts = HipTrainStep(m, B, H, W, operands="bf16" if MODE != "fp32" else "fp32", storage="bf16" if MODE == "bf16s" else "fp32")
x, t = make_batch(B, H, W, 1); x, t = x.cuda(), t.cuda()
for loop in ("table", "python"):
    os.environ["RD_PY_LOOP"] = "1" if loop == "python" else "0"
    for _ in range(5): ts.step(x, t)
    torch.cuda.synchronize()
    c0, t0 = time.thread_time(), time.perf_counter()
    for _ in range(20): ts.step(x, t)
    c1, t1 = time.thread_time(), time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = len(ts._ops)
    print("%s b=%d %dx%d %-6s host issue %.2f ms/step wall, %.2f ms/step CPU (%d ops, %.2f us each); step %.2f ms" % (
        MODE, B, H, W, loop, (t1 - t0) / 20 * 1e3, (c1 - c0) / 20 * 1e3, n, (t1 - t0) / 20 / n * 1e6, (t2 - t0) / 20 * 1e3))
