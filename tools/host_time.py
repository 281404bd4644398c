"""Host enqueue time of one training step (plain launches) against its wall time: the host must stay ahead of the device."""
import sys, time, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import make_batch
args = types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
B, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (16, 450, 800)))     # tiny sizes -> pure host cost per step
MODE = sys.argv[4] if len(sys.argv) > 4 else "fp32"                                       # fp32 | bf16 (operands) | bf16s (storage)
m = create_model(args, [H, W]).cuda()
ts = HipTrainStep(m, B, H, W, operands="bf16" if MODE != "fp32" else "fp32", storage="bf16" if MODE == "bf16s" else "fp32")
x, t = make_batch(B, H, W, 1); x, t = x.cuda(), t.cuda()
for _ in range(5): ts.step(x, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): ts.step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n = len(ts.plan.prep) + len(ts.plan.fwd) + len(ts.plan.bwd)
print(MODE, "host enqueue %.2f ms/step (%d plan ops, %.1f us each); wall %.2f ms/step" % ((t1 - t0) / 20 * 1e3, n, (t1 - t0) / 20 / n * 1e6, (t2 - t0) / 20 * 1e3))
