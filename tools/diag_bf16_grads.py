#!/usr/bin/env python3
"""Per-parameter gradient error of the bf16-operand training step against the fp32 step (one step, lr=1, no momentum / wd, so
the parameter update IS the gradient).  python tools/diag_bf16_grads.py [b h w]"""
import copy, sys, torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch, procedural_fill_
b, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 97, 161)
torch.manual_seed(0)
m0 = ResNet_latefusion(18, "upproj", [h, w], 4, False)
procedural_fill_(m0)
names = [n for n, _ in m0.named_parameters()]
init = [p.detach().clone() for p in m0.parameters()]
x, t = make_batch(b, h, w, 300, ref_pixels=h * w)
g = {}
for ops_ in ("fp32", "bf16"):
    m = copy.deepcopy(m0).cuda()
    ts = HipTrainStep(m, b, h, w, lr=1.0, momentum=0.0, weight_decay=0.0, operands=ops_)
    loss, _ = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    g[ops_] = [i0 - p.detach().cpu() for i0, p in zip(init, m.parameters())]
    print(ops_, "loss", loss.item())
for n, a, c in zip(names, g["fp32"], g["bf16"]):
    e = (a - c).norm().item() / max(a.norm().item(), 1e-30)
    print("%-44s |g| %.3e  rel err %.3e %s" % (n, a.norm().item(), e, "  <<<" if e > 0.05 else ""))
