"""CPU ORACLE (test infrastructure).  Restates the plugin dispatch and the training-step body:

  create_model          <- /root/reference/main.py:118-186 (in-scope archs; others raise like :183)
  compute_loss / train_step <- main.py:416-445 (uncertainty_fixs branch :416-429, default :440-441)
  adjust_learning_rate  <- utils.py:85-89
"""
import torch
import torch.nn as nn

from .criteria import MaskedL1Loss, MaskedMSELoss, SmoothnessLoss
from .models import ResNet_latefusion
from .multistage_model import ResNet_multistage

MULTISTAGE = ("resnet18_multistage", "resnet18_multistage_uncertainty_fixs")


def create_model(args, output_size):
    in_channels = len(args.modality)
    if args.arch == "resnet18_latefusion":
        return ResNet_latefusion(layers=18, decoder=args.decoder, output_size=output_size,
                                 in_channels=in_channels, pretrained=args.pretrained)
    if args.arch == "resnet18_multistage":
        return ResNet_multistage(layers=18, decoder=args.decoder, output_size=output_size, pretrained=args.pretrained)
    if args.arch == "resnet18_multistage_uncertainty_fixs":
        model = ResNet_multistage(layers=18, decoder=args.decoder, output_size=output_size, pretrained=args.pretrained)
        w1 = nn.Parameter(torch.tensor(1.0, dtype=torch.float32), requires_grad=True)
        w2 = nn.Parameter(torch.tensor(1.0, dtype=torch.float32), requires_grad=True)
        model.register_parameter("w_stage1", w1)
        model.register_parameter("w_stage2", w2)
        return model, {"w_stage1": w1, "w_stage2": w2, "w_smooth": 0.1}
    raise ValueError("[Error] Unknown model!!")


def make_criterion(arch, criterion="l1"):
    crit = {"depth": MaskedL1Loss() if criterion == "l1" else MaskedMSELoss()}
    if arch == "resnet18_multistage_uncertainty_fixs":
        crit["smooth"] = SmoothnessLoss()
    return crit


def compute_loss(arch, model, criterion, inputs, target, loss_weights=None):
    """Returns (loss, pred, extras) exactly as the step body composes them."""
    if arch == "resnet18_multistage_uncertainty_fixs":
        out = model(inputs)
        pred1, pred = out["stage1"], out["stage2"]
        d1 = criterion["depth"](pred1, target)
        d2 = criterion["depth"](pred, target)
        sm = criterion["smooth"](pred1, inputs)
        w1, w2 = loss_weights["w_stage1"], loss_weights["w_stage2"]
        loss = torch.exp(-w1) * (d1 + loss_weights["w_smooth"] * sm) + torch.exp(-w2) * d2 + (w1 + w2)
        return loss, pred, {"pred1": pred1, "d1": d1, "d2": d2, "smooth": sm, "out": out}
    if arch in MULTISTAGE:
        out = model(inputs)
        pred1, pred = out["stage1"], out["stage2"]
        d1 = criterion["depth"](pred1, target)
        d2 = criterion["depth"](pred, target)
        return d1 + d2, pred, {"pred1": pred1, "d1": d1, "d2": d2, "out": out}
    pred = model(inputs)
    return criterion["depth"](pred, target), pred, {}


def train_step(arch, model, criterion, optimizer, inputs, target, loss_weights=None):
    loss, pred, extras = compute_loss(arch, model, criterion, inputs, target, loss_weights)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach(), pred.detach(), extras


def adjust_learning_rate(optimizer, epoch, lr_init):
    lr = lr_init * (0.1 ** (epoch // 5))
    for group in optimizer.param_groups:
        group["lr"] = lr
