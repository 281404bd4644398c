#!/usr/bin/env python3
"""Per-tensor distance to the fp64 gradient: fused step (fp32 plan / split plan) and the fp32 CPU oracle, b=2 97x161 (the table behind
tests/test_gpu_margins.py::test_gradients_as_close_to_fp64_as_the_fp32_oracle).   python tools/diag_fp64.py [b h w]"""
import copy
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle.criteria import MaskedL1Loss as OL1  # noqa: E402
from oracle.models import ResNet_latefusion as ORef  # noqa: E402
from radar_depth_amd.main import HipTrainStep  # noqa: E402
from radar_depth_amd.model.models import ResNet_latefusion  # noqa: E402
from radar_depth_amd.synthetic import make_batch, procedural_fill_  # noqa: E402

b, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 97, 161)
torch.manual_seed(0)
o32 = ORef(18, "upproj", [h, w], 4, False)
procedural_fill_(o32)
o32.train()
o64 = copy.deepcopy(o32).double()
x, t = make_batch(b, h, w, 99, ref_pixels=h * w)
y32 = o32(x)
OL1()(y32, t).backward()
y64 = o64(x.double())
OL1()(y64, t.double()).backward()
g32 = [p.grad.double() for p in o32.parameters()]
g64 = [p.grad for p in o64.parameters()]
n64 = np.array([c.norm().item() for c in g64])
e32 = np.array([(a - c).norm().item() for a, c in zip(g32, g64)])
names = [n for n, _ in o32.named_parameters()]
print("forward map: oracle32 vs fp64 max-rel %.3e" % ((y32.double() - y64).abs().max() / y64.abs().max()).item())
res = {}
for operands in ("fp32", "split"):
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(m)
    m = m.cuda().train()
    ts = HipTrainStep(m, b, h, w, operands=operands)
    _, pred = ts.step(x.cuda(), t.cuda())
    torch.cuda.synchronize()
    print("forward map: HIP %s vs fp64 max-rel %.3e" % (operands, ((pred.cpu().double().reshape(y64.shape) - y64).abs().max() / y64.abs().max()).item()))
    g = [m._grad_view(p).detach().cpu().double() for p in m.parameters()]
    res[operands] = np.array([(a - c).norm().item() for a, c in zip(g, g64)])
print("%-50s %10s | %10s %10s %10s   (|| g - g64 || / || g64 ||)" % ("tensor", "|g64|", "oracle32", "HIP fp32", "HIP split"))
order = np.argsort(-(res["split"] / (n64 + 1e-30)))
for k in list(order[:25]) + list(order[-5:]):
    print("%-50s %10.3e | %10.3e %10.3e %10.3e" % (names[k], n64[k], e32[k] / (n64[k] + 1e-30), res["fp32"][k] / (n64[k] + 1e-30), res["split"][k] / (n64[k] + 1e-30)))
print("median: oracle32 %.3e, HIP fp32 %.3e, HIP split %.3e" % (np.median(e32 / (n64 + 1e-30)), np.median(res["fp32"] / (n64 + 1e-30)), np.median(res["split"] / (n64 + 1e-30))))
