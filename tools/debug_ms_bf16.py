import importlib.util, os, sys, types
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import train as otrain
from radar_depth_amd import main as hmain
from radar_depth_amd.synthetic import make_batch, procedural_fill_
spec = importlib.util.spec_from_file_location("_bf", "tests/test_gpu_bf16.py"); bf = importlib.util.module_from_spec(spec); spec.loader.exec_module(bf)
b, h, w = 2, 97, 161
args = types.SimpleNamespace(arch="resnet18_multistage_uncertainty_fixs", decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
hm, hw_ = hmain.create_model(args, [h, w]); om, ow = otrain.create_model(args, [h, w])
procedural_fill_(hm); procedural_fill_(om)
hm = hm.cuda().train(); om.train()
print("emulated convs", bf._emulate_bf16_operands(om))
x, t = make_batch(b, h, w, 600, ref_pixels=h * w)
crit = otrain.make_criterion(args.arch)
lo, po, ex = otrain.compute_loss(args.arch, om, crit, x, t, ow)
def rel(a, c): return ((a.cpu() - c).abs().max() / c.abs().max()).item()
mp = hm._plans(b, h, w, True, bf16=True)
mp.run_forward(x.cuda()); torch.cuda.synchronize()
print("fwd only: stage1 %.3e stage2 %.3e kept %.3e" % (rel(mp.p1.pred, ex["pred1"].detach()), rel(mp.p2.pred, po.detach()), rel(mp.kept, ex["out"]["radar_filtered"])))
# feed stage 2 of the oracle with the HIP stage-1 prediction to separate the coupling from stage 2's own arithmetic
with torch.no_grad():
    p1h = mp.p1.pred.cpu()
    kept, mask = om.filter_layer(x[:, 3:4], p1h)
    o2 = om.stage2(torch.cat((x[:, :3], kept, p1h), 1))
print("stage2 vs oracle stage2 fed with HIP stage-1 output: %.3e" % rel(mp.p2.pred, o2))
for name in ("conv1_depth", "maxpool_depth", "layer1_depth.0", "layer4_depth.1", "layer4.1", "bn_fusion", "bn2", "decoder.layer1", "decoder.layer4"):
    pass
mp32 = hm._plans(b, h, w, True, bf16=False)
mp32.run_forward(x.cuda()); torch.cuda.synchronize()
print("fp32 plan vs bf16-emulated oracle: stage1 %.3e stage2 %.3e" % (rel(mp32.p1.pred, ex["pred1"].detach()), rel(mp32.p2.pred, po.detach())))
