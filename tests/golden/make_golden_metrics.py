#!/usr/bin/env python3
"""Golden vectors for Result.evaluate / AverageMeter from the REAL reference (evaluation/metrics.py imports only torch,
math, numpy -- no shim needed).  Authoring container only:  python tests/golden/make_golden_metrics.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from evaluation.metrics import AverageMeter, Result  # noqa: E402

g = torch.Generator().manual_seed(21)
out = torch.rand(2, 1, 37, 53, generator=g) * 70 + 0.5
tgt = torch.where(torch.rand(2, 1, 37, 53, generator=g) < 0.3, torch.rand(2, 1, 37, 53, generator=g) * 80 + 0.1, torch.zeros(2, 1, 37, 53))
names = ("irmse", "imae", "mse", "rmse", "mae", "absrel", "lg10", "delta1", "delta2", "delta3")
r = Result()
r.evaluate(out, tgt)
r2 = Result()
r2.evaluate(out * 1.1 + 0.3, tgt)
m = AverageMeter()
m.update(r, 0.5, 0.1, 2)
m.update(r2, 0.7, 0.2, 3)
a = m.average()
np.savez_compressed(os.path.join(HERE, "metrics.npz"), out=out.numpy(), target=tgt.numpy(), names=np.array(names),
                    r1=np.array([getattr(r, n) for n in names]), r2=np.array([getattr(r2, n) for n in names]),
                    avg=np.array([getattr(a, n) for n in names] + [a.gpu_time, a.data_time]))
print("wrote metrics.npz", [round(getattr(r, n), 5) for n in names])
