"""bf16-storage smoke: one latefusion step at 2x97x161 vs the fp32 HIP step (loose), then NaN checks at b=16 450x800."""
import sys, torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep
from radar_depth_amd.model.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch, procedural_fill_
b, h, w = 2, 97, 161
res = {}
for st in ("fp32", "bf16"):
    torch.manual_seed(0)
    m = ResNet_latefusion(18, "upproj", [h, w], 4, False); procedural_fill_(m); m = m.cuda()
    ts = HipTrainStep(m, b, h, w, storage=st)
    losses = []
    for it in range(3):
        x, t = make_batch(b, h, w, 300 + it, ref_pixels=h * w)
        loss, pred = ts.step(x.cuda(), t.cuda()); torch.cuda.synchronize()
        losses.append(loss.item())
    res[st] = (losses, pred.clone())
    print(st, losses, "finite", all(torch.isfinite(p).all().item() for p in m.parameters()))
print("pred rel diff", ((res["bf16"][1] - res["fp32"][1]).abs().max() / res["fp32"][1].abs().max()).item())
