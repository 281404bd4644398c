"""MI355X-native counterpart of the reference's evaluation/criteria_new.py (hot-path subset):
MaskedL1Loss (:44-54), MaskedMSELoss (:31-41) and SmoothnessLoss (:8-28) as HIP reductions behind torch.autograd.  Same call
signatures; losses are 0-dim CUDA tensors; an all-invalid target gives NaN exactly like the reference."""
import ctypes as C

import torch
import torch.nn as nn

from .._lib import check, current_stream, lib, ptr


def _f64(n, dev):
    return torch.zeros(n, dtype=torch.float64, device=dev)


class _MaskedL1Fn(torch.autograd.Function):
    SUMS, BWD = "rd_masked_l1_sums", "rd_masked_l1_bwd"

    @classmethod
    def forward(cls, ctx, pred, target):
        L = lib()
        pred = pred.contiguous()
        target = target.contiguous()
        n = pred.numel()
        tiles = L.rd_loss_tiles(C.c_int64(n))
        ws = _f64(2 * tiles, pred.device)
        sums = _f64(2, pred.device)
        check(getattr(L, cls.SUMS)(ptr(pred), ptr(target), C.c_int64(n), ptr(ws), ptr(sums), current_stream()), cls.SUMS)
        ctx.save_for_backward(pred, target, sums)
        return (sums[0] / sums[1]).float()

    @classmethod
    def backward(cls, ctx, gout):
        pred, target, sums = ctx.saved_tensors
        dpred = torch.empty_like(pred)
        coef = gout.reshape(1).float().contiguous()
        check(getattr(lib(), cls.BWD)(ptr(pred), ptr(target), C.c_int64(pred.numel()), ptr(sums), ptr(coef), ptr(dpred), 0,
                                      current_stream()), cls.BWD)
        return dpred, None


class _MaskedL2Fn(_MaskedL1Fn):
    SUMS, BWD = "rd_masked_l2_sums", "rd_masked_l2_bwd"


class MaskedMSELoss(nn.Module):
    """`-c l2` (main.py:294-305): mean of (target - pred)^2 over target > 0, NaN on an all-invalid target."""

    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        if not pred.is_cuda:
            raise RuntimeError("radar_depth_amd losses run on MI355X only (HIP kernels)")
        self.loss = _MaskedL2Fn.apply(pred.float(), target.float())
        return self.loss


class MaskedL1Loss(nn.Module):
    def forward(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        if not pred.is_cuda:
            raise RuntimeError("radar_depth_amd losses run on MI355X only (HIP kernels)")
        self.loss = _MaskedL1Fn.apply(pred.float(), target.float())
        return self.loss


class _SmoothFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, image):
        L = lib()
        pred = pred.contiguous()
        image = image.contiguous()
        n, _, h, w = pred.shape
        c = image.shape[1]
        nfl = int(L.rd_smooth_workspace_floats(n, h, w))
        ws = torch.empty((nfl + 1) // 2, dtype=torch.float64, device=pred.device)
        out = _f64(1, pred.device)
        check(L.rd_smooth_fwd(ptr(pred), ptr(image), n, c, h, w, ptr(ws), ptr(out), current_stream()), "rd_smooth_fwd")
        ctx.ws, ctx.shape = ws, (n, h, w)
        return out[0].float()

    @staticmethod
    def backward(ctx, gout):
        n, h, w = ctx.shape
        dpred = torch.empty((n, 1, h, w), dtype=torch.float32, device=gout.device)
        coef = gout.reshape(1).float().contiguous()
        check(lib().rd_smooth_bwd(n, h, w, ptr(ctx.ws), ptr(coef), ptr(dpred), 0, current_stream()), "rd_smooth_bwd")
        return dpred, None


class SmoothnessLoss(nn.Module):
    def forward(self, pred_depth, image):
        if not pred_depth.is_cuda:
            raise RuntimeError("radar_depth_amd losses run on MI355X only (HIP kernels)")
        assert pred_depth.shape[1] == 1 and pred_depth.shape[2:] == image.shape[2:]
        return _SmoothFn.apply(pred_depth.float(), image.float())
