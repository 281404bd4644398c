"""CPU ORACLE (test infrastructure).  Restates /root/reference/evaluation/criteria_new.py:
SmoothnessLoss :8-28, MaskedMSELoss :31-41, MaskedL1Loss :44-54."""
import torch
import torch.nn as nn


class SmoothnessLoss(nn.Module):
    def forward(self, pred_depth, image):
        mean = pred_depth.mean(2, True).mean(3, True)
        d = pred_depth / (mean + 1e-7)
        gx = (d[:, :, :, :-1] - d[:, :, :, 1:]).abs()
        gy = (d[:, :, :-1, :] - d[:, :, 1:, :]).abs()
        ix = (image[:, :, :, :-1] - image[:, :, :, 1:]).abs().mean(1, keepdim=True)
        iy = (image[:, :, :-1, :] - image[:, :, 1:, :]).abs().mean(1, keepdim=True)
        return (gx * torch.exp(-ix)).mean() + (gy * torch.exp(-iy)).mean()


class _Masked(nn.Module):
    def _diff(self, pred, target):
        assert pred.dim() == target.dim(), "inconsistent dimensions"
        valid = (target > 0).detach()
        return (target - pred)[valid]


class MaskedMSELoss(_Masked):
    def forward(self, pred, target):
        self.loss = (self._diff(pred, target) ** 2).mean()
        return self.loss


class MaskedL1Loss(_Masked):
    def forward(self, pred, target):
        self.loss = self._diff(pred, target).abs().mean()
        return self.loss
