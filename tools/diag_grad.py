"""Conditioning check: per-parameter gradient error of (a) the fp32 CPU oracle and (b) the HIP path, both
measured against the fp64 CPU oracle on the same batch."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle.criteria import MaskedL1Loss as OL1
from oracle.models import ResNet_latefusion as ORef
from radar_depth_amd.evaluation.criteria_new import MaskedL1Loss
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch, procedural_fill_

b, h, w = 2, 97, 161
x, t = make_batch(b, h, w, 4321, ref_pixels=h * w)


def oracle(dtype):
    torch.manual_seed(0)
    o = ORef(18, "upproj", [h, w], 4, False)
    procedural_fill_(o)
    o = o.to(dtype).train()
    y = o(x.to(dtype))
    loss = OL1()(y, t.to(dtype))
    loss.backward()
    return {n: p.grad.double().numpy() for n, p in o.named_parameters()}, y.detach().double().numpy()


g64, y64 = oracle(torch.float64)
g32, y32 = oracle(torch.float32)
m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
procedural_fill_(m)
m = m.cuda().train()
y = m(x.cuda())
loss = MaskedL1Loss()(y, t.cuda())
loss.backward()
gh = {n: p.grad.double().cpu().numpy() for n, p in m.named_parameters()}
yh = y.detach().double().cpu().numpy()
print("fwd err vs fp64: oracle32 %.2e  hip %.2e" % (np.abs(y32 - y64).max() / np.abs(y64).max(), np.abs(yh - y64).max() / np.abs(y64).max()))
rows = []
for n in g64:
    s = np.abs(g64[n]).max() + 1e-30
    rows.append((np.abs(gh[n] - g64[n]).max() / s, np.abs(g32[n] - g64[n]).max() / s, n))
rows.sort(reverse=True)
print("%-45s %10s %10s" % ("param", "hip", "oracle32"))
for eh, eo, n in rows[:25]:
    print("%-45s %10.2e %10.2e" % (n, eh, eo))
print("median hip %.2e oracle32 %.2e" % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
order = ["conv3.weight"] + [k for i in (4, 3, 2, 1) for k in g64 if k.startswith("decoder.layer%d" % i)] + \
        ["bn2.weight", "bn2.bias", "conv2.weight", "bn_fusion.weight", "conv_fusion.weight", "layer4.1.conv2.weight", "layer4.1.bn2.weight",
         "layer4_depth.1.conv2.weight", "layer4_depth.1.bn2.weight"]
print("---- backward order")
d = {n: (a, b) for a, b, n in rows}
for n in order:
    print("%-45s %10.2e %10.2e" % (n, d[n][0], d[n][1]))
