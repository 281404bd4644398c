#!/usr/bin/env python3
"""Where the three streams of the plan are at the forks / joins of an UN-PROFILED step (RD_TAIL_EVENTS=1: timing events recorded from the
op list).  Under rocprofv3 the host falls behind the device and the order in which it issues the streams' ops shapes the trace; here the
host runs ahead as in bench.py.    python tools/tail_probe.py [config=2] [steps=12]"""
import os
import sys

os.environ["RD_TAIL_EVENTS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bench import CONFIGS  # noqa: E402
from radar_depth_amd.main import HipTrainStep, create_model  # noqa: E402
from radar_depth_amd.synthetic import make_batch  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
arch, b, h, w, storage = CONFIGS[cfg]
torch.manual_seed(0)


import types  # noqa: E402

made = create_model(types.SimpleNamespace(arch=arch, decoder="upproj", modality="rgbd", pretrained=False), [h, w])
net, loss_weights = made if isinstance(made, tuple) else (made, None)
net = net.cuda()
x, t = make_batch(b, h, w, 7)
x, t = x.cuda(), t.cuda()
ts = HipTrainStep(net, b, h, w, lr=0.01, momentum=0.9, weight_decay=1e-4, loss_weights=loss_weights, use_graph=False, storage=storage)
for _ in range(steps):
    ts.step(x, t)
torch.cuda.synchronize()
for plan in ts.plans:
    pr = {n: ev for n, k, ev in plan.probes}
    base = pr["fwd_begin"]
    print("# plan %s: offsets from fwd_begin (ms) in the last of %d steps" % (type(plan).__name__, steps))
    for n, k, ev in plan.probes:
        print("  %-28s stream %d  %8.3f" % (n, k, base.elapsed_time(ev)))
