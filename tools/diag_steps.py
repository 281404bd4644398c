"""Per-step parameter drift of (a) the fused HipTrainStep (eager / graph) and (b) the autograd-compatible HIP path,
both against the fp32 CPU oracle, on the small geometry."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle.criteria import MaskedL1Loss as OL1
from oracle.models import ResNet_latefusion as ORef
from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss
from radar_depth_amd.main import HipTrainStep
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch, procedural_fill_

b, h, w = 2, 97, 161
names = ("conv1_depth.weight", "conv1.weight", "conv3.weight", "layer4.1.conv2.weight", "bn1_depth.weight", "layer1_depth.0.conv1.weight")


def mk(cls):
    torch.manual_seed(0)
    m = cls(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    return m


o = mk(ORef).train()
oo = torch.optim.SGD(o.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
variants = {}
for tag, graph in (("fused-eager", False), ("fused-graph", True)):
    m = mk(ResNet_latefusion).cuda()
    variants[tag] = (m, HipTrainStep(m, b, h, w, use_graph=graph))
mc = mk(ResNet_latefusion).cuda().train()
oc = torch.optim.SGD(mc.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
for it in range(4):
    x, t = make_batch(b, h, w, 99 + it, ref_pixels=h * w)
    lo = OL1()(o(x), t); oo.zero_grad(); lo.backward()
    og = {n: p.grad.clone() for n, p in o.named_parameters()}
    oo.step()
    line = "step %d oracle loss %.5f |" % (it, lo.item())
    for tag, (m, ts) in variants.items():
        l, _ = ts.step(x.cuda(), t.cuda()); torch.cuda.synchronize()
        line += " %s %.5f" % (tag, l.item())
    lc = MaskedL1Loss()(mc(x.cuda()), t.cuda()); oc.zero_grad(); lc.backward()
    gc = {n: p.grad.clone().cpu() for n, p in mc.named_parameters()}
    oc.step()
    print(line + " compat %.5f" % lc.item())
    od = dict(o.named_parameters())
    for n in names:
        r = ["%s %.1e" % (tag, ((dict(m.named_parameters())[n].detach().cpu() - od[n].detach()).norm() / od[n].detach().norm()).item())
             for tag, (m, _) in variants.items()]
        r.append("compat %.1e" % ((dict(mc.named_parameters())[n].detach().cpu() - od[n].detach()).norm() / od[n].detach().norm()).item())
        r.append("grad(compat) %.1e" % ((gc[n] - og[n]).norm() / og[n].norm()).item())
        print("    %-28s %s" % (n, "  ".join(r)))
