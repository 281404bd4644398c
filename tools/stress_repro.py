"""Race hunt: two identically initialised training steps (latefusion at several geometries, then multistage) must stay
bit-identical while ~650 kernels run on three streams.   python tools/stress_repro.py [fp32|bf16|bf16s]"""
import sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd.main import HipTrainStep, create_model
from radar_depth_amd.synthetic import make_batch, procedural_fill_


def build(arch, h, w):
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    made = create_model(args, [h, w])
    m, lw = made if isinstance(made, tuple) else (made, None)
    procedural_fill_(m)
    return m.cuda(), lw


MODE = ([a_ for a_ in sys.argv[1:] if not a_.startswith("--")] or ["fp32"])[0]          # fp32 | bf16 (operands) | bf16s (bf16 storage)
assert MODE in ("fp32", "bf16", "bf16s"), MODE
OPERANDS, STORAGE = ("bf16" if MODE != "fp32" else "fp32"), ("bf16" if MODE == "bf16s" else "fp32")
print("mode:", MODE)
bad = 0
CASES = [("resnet18_latefusion", 16, 450, 800, 6), ("resnet18_latefusion", 3, 225, 401, 6), ("resnet18_latefusion", 1, 450, 800, 6),
         ("resnet18_latefusion", 5, 97, 161, 8), ("resnet18_multistage_uncertainty_fixs", 4, 225, 400, 5),
         ("resnet18_multistage_uncertainty_fixs", 8, 450, 800, 3)]
if "--long" in sys.argv:       # the geometry that exposed the LDS race of round 2 (b=2, 450x800) and more steps everywhere
    CASES = [(a_, b_, h_, w_, 4 * s_) for a_, b_, h_, w_, s_ in CASES] + [("resnet18_multistage_uncertainty_fixs", 2, 450, 800, 30),
                                                                          ("resnet18_latefusion", 2, 450, 800, 30),
                                                                          ("resnet18_latefusion", 2, 900, 1600, 10)]
for arch, b, h, w, steps in CASES:
    (m1, lw1), (m2, lw2) = build(arch, h, w), build(arch, h, w)
    t1 = HipTrainStep(m1, b, h, w, loss_weights=lw1, operands=OPERANDS, storage=STORAGE)
    t2 = HipTrainStep(m2, b, h, w, loss_weights=lw2, operands=OPERANDS, storage=STORAGE)
    ok = True
    for it in range(steps):
        x, t = make_batch(b, h, w, 4000 + it, ref_pixels=h * w)
        l1, _ = t1.step(x.cuda(), t.cuda())
        l2, _ = t2.step(x.cuda(), t.cuda())
        torch.cuda.synchronize()
        ok = ok and l1.item() == l2.item()
    ok = ok and all(torch.equal(p, q) and bool(torch.isfinite(p).all()) for p, q in zip(m1.parameters(), m2.parameters()))
    print("%-40s b=%d %dx%d: %s (final loss %.6f)" % (arch, b, h, w, "bit-identical" if ok else "MISMATCH", l1.item()))
    bad += not ok
    del t1, t2, m1, m2
    torch.cuda.empty_cache()
print("%d mismatching configurations" % bad)
