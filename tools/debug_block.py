"""Stage-by-stage comparison of one BasicBlock run through ModulePlan against torch on the GPU box (debugging aid)."""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radar_depth_amd.engine import ModulePlan
from radar_depth_amd.model.models import ArenaOwner, BasicBlock, _conv
from radar_depth_amd.synthetic import procedural_fill_

want = np.load("tests/golden/basic_block.npz")
tag, cin, cout, stride = "ds", 32, 64, 2
down = torch.nn.Sequential(_conv(cin, cout, 1, stride, pad=0), torch.nn.BatchNorm2d(cout))
mod = BasicBlock(cin, cout, stride, down)
procedural_fill_(mod)

class Single(ArenaOwner, torch.nn.Module):
    def __init__(self, inner):
        super().__init__()
        self.mod = inner
owner = Single(mod).cuda()
x = torch.tensor(want[tag + "/x"]).cuda()
n, c, h, w = x.shape
plan = ModulePlan(owner, owner.mod, "block", n, h, w, c)
y, dx = plan.run(x, torch.tensor(want[tag + "/gy"]).cuda())
m = owner.mod
def bn(t, b):
    return F.batch_norm(t, None, None, b.weight, b.bias, True, 0.1, 1e-5)
r1 = F.conv2d(x, m.conv1.weight, None, stride, 1)
y1 = F.relu(bn(r1, m.bn1))
r2 = F.conv2d(y1, m.conv2.weight, None, 1, 1)
rd = F.conv2d(x, m.downsample[0].weight, None, stride, 0)
yy = F.relu(bn(r2, m.bn2) + bn(rd, m.downsample[1]))
def cmp(name, ref):
    got = plan.taps[name].view().permute(0, 3, 1, 2)
    d = (got - ref).abs()
    print("%-20s max err %.3e of %.3e; bad elems %d / %d" % (name, d.max().item(), ref.abs().max().item(), (d > 1e-4 * ref.abs().max()).sum().item(), d.numel()))
    if d.max() > 1e-3:
        bad = (d > 1e-3).nonzero()
        print("   first bad idx (n,c,h,w):", bad[:6].tolist(), " last:", bad[-3:].tolist())
for name, ref in (("m.conv1", r1), ("m.relu1", y1), ("m.conv2", r2), ("m.downsample.0", rd), ("m", yy)):
    cmp(name, ref)
print("vs golden y:", (y.cpu() - torch.tensor(want[tag + "/y"])).abs().max().item(), " torch-vs-golden:", (yy.cpu() - torch.tensor(want[tag + "/y"])).abs().max().item())
