"""MI355X-native counterpart of the reference's evaluation/metrics.py:
Result.evaluate (:34-58) as one fused masked reduction on the device + a single 80-byte readback (the reference issues
about a dozen blocking float() conversions per call), and the AverageMeter (:179-216).  Same attribute names."""
import ctypes as C
import math

import numpy as np
import torch

from .._lib import check, current_stream, lib, ptr


_ERRORS = ("irmse", "imae", "mse", "rmse", "mae", "absrel", "lg10")     # lower is better -> worst = +inf
_SCORES = ("delta1", "delta2", "delta3")                                # higher is better -> worst = 0
_TIMES = ("gpu_time", "data_time")
_METRICS = _ERRORS + _SCORES


class Result(object):
    """Record of one evaluation (same attribute names and `update` argument order as evaluation/metrics.py:10-31; picklable
    under the same module path so reference checkpoints' `best_result` round-trips)."""

    def __init__(self):
        self._assign(dict.fromkeys(_METRICS + _TIMES, 0))

    def _assign(self, values):
        for key, val in values.items():
            setattr(self, key, val)

    def set_to_worst(self):
        self._assign({**dict.fromkeys(_ERRORS, np.inf), **dict.fromkeys(_SCORES + _TIMES, 0)})

    def update(self, irmse, imae, mse, rmse, mae, absrel, lg10, delta1, delta2, delta3, gpu_time, data_time):
        self._assign(dict(zip(_METRICS + _TIMES, (irmse, imae, mse, rmse, mae, absrel, lg10, delta1, delta2, delta3,
                                                  gpu_time, data_time))))

    def evaluate(self, output, target):
        if not output.is_cuda:
            raise RuntimeError("radar_depth_amd metrics run on MI355X only (HIP kernels)")
        L = lib()
        output = output.contiguous().float()
        target = target.contiguous().float()
        n = output.numel()
        tiles = L.rd_loss_tiles(C.c_int64(n))
        ws = torch.empty(10 * tiles, dtype=torch.float64, device=output.device)
        sums = torch.empty(10, dtype=torch.float64, device=output.device)
        check(L.rd_depth_metrics(ptr(output), ptr(target), C.c_int64(n), ptr(ws), ptr(sums), current_stream()), "rd_depth_metrics")
        s = sums.cpu().numpy()          # the only host synchronisation
        cnt = s[0]
        mean = (lambda v: float(v / cnt)) if cnt > 0 else (lambda v: float("nan"))
        self.mse = mean(s[1])
        self.rmse = math.sqrt(self.mse) if cnt > 0 else float("nan")
        self.mae = mean(s[2])
        self.lg10 = mean(s[3])
        self.absrel = mean(s[4])
        self.delta1, self.delta2, self.delta3 = mean(s[5]), mean(s[6]), mean(s[7])
        self.data_time = 0
        self.gpu_time = 0
        self.irmse = math.sqrt(mean(s[8])) if cnt > 0 else float("nan")
        self.imae = mean(s[9])


class AverageMeter(object):
    _FIELDS = _METRICS

    def __init__(self):
        self.reset()

    def reset(self):
        self.count = 0.0
        for f in self._FIELDS:
            setattr(self, "sum_" + f, 0)
        self.sum_data_time, self.sum_gpu_time = 0, 0

    def update(self, result, gpu_time, data_time, n=1):
        self.count += n
        for f in self._FIELDS:
            setattr(self, "sum_" + f, getattr(self, "sum_" + f) + n * getattr(result, f))
        self.sum_data_time += n * data_time
        self.sum_gpu_time += n * gpu_time

    def average(self):
        avg = Result()
        c = self.count
        avg.update(self.sum_irmse / c, self.sum_imae / c, self.sum_mse / c, self.sum_rmse / c, self.sum_mae / c,
                   self.sum_absrel / c, self.sum_lg10 / c, self.sum_delta1 / c, self.sum_delta2 / c, self.sum_delta3 / c,
                   self.sum_gpu_time / c, self.sum_data_time / c)
        return avg
