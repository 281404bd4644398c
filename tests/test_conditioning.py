"""Documents (on CPU, with the oracle alone) why end-to-end multi-step comparisons use losses / norms and loose
element-wise tolerances: the reference network's gradients are extremely sensitive to tiny weight perturbations."""
import copy

import torch

from oracle.criteria import MaskedL1Loss
from oracle.models import ResNet_latefusion
from radar_depth_amd.synthetic import make_batch, procedural_fill_


def test_oracle_gradient_sensitivity():
    b, h, w = 2, 97, 161
    torch.manual_seed(0)
    a = ResNet_latefusion(18, "upproj", [h, w], 4, False)
    procedural_fill_(a)
    a.train()
    opt = torch.optim.SGD(a.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    x, t = make_batch(b, h, w, 99, ref_pixels=h * w)
    loss = MaskedL1Loss()(a(x), t)
    opt.zero_grad()
    loss.backward()
    opt.step()
    pert = copy.deepcopy(a)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for n, p in pert.named_parameters():
            if n in ("conv1.weight", "conv1_depth.weight"):
                p.mul_(1 + 5e-4 * torch.randn(p.shape, generator=g))
    x, t = make_batch(b, h, w, 100, ref_pixels=h * w)
    la = MaskedL1Loss()(a(x), t)
    a.zero_grad()
    la.backward()
    lb = MaskedL1Loss()(pert(x), t)
    pert.zero_grad()
    lb.backward()
    assert abs(la.item() - lb.item()) / la.item() < 1e-3          # the loss barely moves ...
    ga, gb = a.layer4[1].conv2.weight.grad, pert.layer4[1].conv2.weight.grad
    assert ((ga - gb).norm() / ga.norm()).item() > 1e-2            # ... while deep gradients move by >1 % (measured ~16 %)
