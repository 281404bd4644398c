"""Race hunt for the autograd (eager plan) path: the same multistage forward + backward repeated in one process must be
bit-identical every time (the radar filter turns a one-ulp difference of stage 1 into a percent-level change of stage 2, so the
mask and stage-2 map are very sensitive detectors).   python tools/stress_eager.py [reps]"""
import sys, types
import torch
sys.path.insert(0, ".")
from radar_depth_amd.main import create_model
from radar_depth_amd.synthetic import make_batch, procedural_fill_

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
only = int(sys.argv[2]) if len(sys.argv) > 2 else -1
bad = 0
CASES = (("resnet18_multistage_uncertainty_fixs", 2, 450, 800), ("resnet18_latefusion", 3, 225, 401),
         ("resnet18_multistage_uncertainty_fixs", 2, 97, 161))
for arch, b, h, w in (CASES if only < 0 else CASES[only:only + 1]):
    args = types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False)
    torch.manual_seed(0)
    made = create_model(args, [h, w])
    m = (made[0] if isinstance(made, tuple) else made)
    procedural_fill_(m)
    m = m.cuda().train()
    x, t = make_batch(b, h, w, 77, ref_pixels=h * w)
    x = x.cuda()
    ref = None
    for it in range(reps):
        m.zero_grad(set_to_none=True)
        o = m(x)
        keys = [k for k in sorted(o) if torch.is_tensor(o[k])] if isinstance(o, dict) else ["out"]
        outs = [o[k] for k in keys] if isinstance(o, dict) else [o]
        loss = sum(v.float().mean() for v in outs if v.dtype.is_floating_point)
        loss.backward()
        torch.cuda.synchronize()
        cur = [v.detach().clone() for v in outs] + [p.grad.detach().clone() for p in m.parameters() if p.grad is not None]
        if ref is None:
            ref = cur
            prev = cur
        else:
            diff = [i for i, (a_, b_) in enumerate(zip(ref, cur)) if not torch.equal(a_, b_)]
            if diff:
                bad += 1
                names = keys + ["grad:" + n for n, p in m.named_parameters() if p.grad is not None]
                d0 = diff[0]
                md = (ref[d0].float() - cur[d0].float()).abs().max().item()
                same_prev = all(torch.equal(a_, b_) for a_, b_ in zip(prev, cur))
                print("%s rep %d: %d tensors differ from rep 0 (first: %s, max abs diff %.3e; outputs differing: %s); identical to previous rep: %s"
                      % (arch, it, len(diff), names[d0], md, [names[i] for i in diff if i < len(keys)], same_prev))
        prev = cur
        # shuffle the allocator between repetitions
        junk = [torch.empty(int(1e6 * (1 + (it * 7 + k) % 5)), device="cuda") for k in range(3)]
        del junk
    print("%s b=%d %dx%d: %d repetitions compared" % (arch, b, h, w, reps))
    del m
    torch.cuda.empty_cache()
print("%d mismatching repetitions" % bad)
