"""Find the first intermediate tensor that differs between two identical runs of the plan (race hunt).
   python tools/diag_repro.py B H W [reps]"""
import sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import make_batch, procedural_fill_
B, H, W = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
args = types.SimpleNamespace(arch="resnet18_latefusion", decoder="upproj", modality="rgbd", pretrained=False)
torch.manual_seed(0)
m = create_model(args, [H, W]); procedural_fill_(m); m = m.cuda()
ts = HipTrainStep(m, B, H, W, lr=0.0)          # lr 0: parameters stay put, every step must reproduce the first one bit for bit
x, t = make_batch(B, H, W, 4000, ref_pixels=H * W); x, t = x.cuda(), t.cuda()
plan = ts.plan
ref = None
arena0 = ts.st["arena"].clone()
bufs0 = [b.clone() for b in m.buffers()]
for rep in range(reps):
    ts.st["arena"].copy_(arena0); ts.st["mom"].zero_()          # every rep starts from the same state
    for b, b0 in zip(m.buffers(), bufs0): b.copy_(b0)
    torch.cuda.synchronize()
    ts.step(x, t)
    torch.cuda.synchronize()
    snap = {k: v.t.clone() for k, v in plan.taps.items()}
    snap["__grads"] = ts.st["grads"].clone()
    if ref is None:
        ref = snap
        continue
    diff = [(k, (snap[k] - ref[k]).abs().max().item()) for k in ref if not torch.equal(snap[k], ref[k])]
    print("rep %d: %d of %d tensors differ" % (rep, len(diff), len(ref)), diff[:6])
    if "__grads" in dict(diff):
        gd = (snap["__grads"] - ref["__grads"])
        bad = torch.nonzero((gd != 0) | torch.isnan(gd)).flatten()
        offs = plan.m._param_offsets() if hasattr(plan.m, "_param_offsets") else None
        print("   grads differ at %d elements, first %d last %d" % (bad.numel(), bad[0].item(), bad[-1].item()))
        lo = 0
        for n, p_ in m.named_parameters():
            hi = lo + p_.numel()
            c = ((bad >= p_.data_ptr() // 4 - ts.st["arena"].data_ptr() // 4) & (bad < p_.data_ptr() // 4 - ts.st["arena"].data_ptr() // 4 + p_.numel())).sum().item()
            if c: print("      %s: %d elements" % (n, c))
