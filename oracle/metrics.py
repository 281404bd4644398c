"""CPU ORACLE (test infrastructure).  Restates /root/reference/evaluation/metrics.py:34-58 (Result.evaluate) as a
function returning the ten metrics in the reference's order (irmse, imae, mse, rmse, mae, absrel, lg10, delta1..3)."""
import math

import torch


def evaluate(output, target):
    valid = target > 0
    o, t = output[valid], target[valid]
    ad = (o - t).abs()
    mse = float((ad ** 2).mean())
    lg10 = float((torch.log(o) / math.log(10) - torch.log(t) / math.log(10)).abs().mean())
    ratio = torch.max(o / t, t / o)
    inv = (1 / o - 1 / t).abs()
    return [math.sqrt(float((inv ** 2).mean())), float(inv.mean()), mse, math.sqrt(mse), float(ad.mean()), float((ad / t).mean()),
            lg10, float((ratio < 1.25).float().mean()), float((ratio < 1.25 ** 2).float().mean()),
            float((ratio < 1.25 ** 3).float().mean())]
