"""Initialiser pin: the reference's freshly constructed ResNet_latefusion / ResNet_multistage
(/root/reference/model/models.py:30-72 applied at :541-542, :561-562, :591-594, :622-623; multistage_model.py:22-61) were
summarised per state_dict tensor -- n, mean, std, abs-max, kurtosis -- by tests/golden/make_golden.py (init_stats.npz).
Both the product constructors (radar_depth_amd.model) and the oracle's (oracle.models) must draw every tensor from the same
law: exact for the constant fills (BatchNorm 1/0, running stats), statistical for the random ones (normal vs uniform is
told apart by the kurtosis: 3.0 vs 1.8 -- that is what pins "conv1_depth / conv_fusion keep PyTorch's default
kaiming_uniform(a=sqrt 5)" and "the RGB stem is N(0, sqrt(2/(49*64)))" after its double initialisation).

The torch RNG stream cannot be matched draw for draw (the stand-in torchvision resnet18 used to import the reference
consumes it differently from the real one), hence moments with sample-size-aware tolerances:
    |std/std_ref - 1| < 6/sqrt(n) + 1e-3,   |mean - mean_ref| < 6 std_ref sqrt(2/n),   kurtosis class equal for n >= 700."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "init_stats.npz")


def _moments(model):
    names, rows = [], []
    for k, v in model.state_dict().items():
        if v.dim() == 0 and not v.is_floating_point():
            continue
        d = v.detach().double().flatten()
        c = d - d.mean()
        m2 = (c ** 2).mean().item()
        kurt = ((c ** 4).mean().item() / (m2 * m2)) if m2 > 0 else 0.0
        names.append(k)
        rows.append([d.numel(), d.mean().item(), d.std(unbiased=False).item() if d.numel() > 1 else 0.0, d.abs().max().item(), kurt])
    return names, np.array(rows)


def _compare(names, rows, want_names, want_rows):
    assert names == [str(n) for n in want_names]
    n_random = n_uniform = 0
    for name, got, ref in zip(names, rows, want_rows):
        n, mean, std, amax, kurt = ref
        assert got[0] == n, name
        if std == 0.0:                      # constant fill: exact
            assert got[1] == mean and got[2] == 0.0 and got[3] == amax, (name, got, ref)
            continue
        n_random += 1
        assert abs(got[2] / std - 1.0) < 6.0 / np.sqrt(n) + 1e-3, (name, got[2], std)
        assert abs(got[1] - mean) < 6.0 * std * np.sqrt(2.0 / n), (name, got[1], mean)
        if n >= 700:
            uniform_ref = kurt < 2.4
            assert (got[4] < 2.4) == uniform_ref, (name, got[4], kurt)
            if uniform_ref:                 # bounded support: |w| <= 1/sqrt(fan_in) for kaiming_uniform(a=sqrt 5)
                n_uniform += 1
                assert abs(got[3] / amax - 1.0) < 0.02, (name, got[3], amax)
    return n_random, n_uniform


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_latefusion_initialisers_match_reference(impl):
    want = np.load(GOLD)
    if impl == "product":
        from radar_depth_amd.model.models import ResNet_latefusion
    else:
        from oracle.models import ResNet_latefusion
    torch.manual_seed(99)
    names, rows = _moments(ResNet_latefusion(18, "upproj", [450, 800], 4, False))
    n_random, n_uniform = _compare(names, rows, want["lf_names"], want["lf_rows"])
    # 55 convolutions are random; exactly two of the large ones keep the default uniform init (conv1_depth, conv_fusion)
    assert n_random == 55 and n_uniform == 2
    ref = dict(zip([str(n) for n in want["lf_names"]], want["lf_rows"]))
    assert ref["conv1_depth.weight"][4] < 2.4 and ref["conv_fusion.weight"][4] < 2.4 and ref["conv1.weight"][4] > 2.5
    # the RGB stem after its double init (models.py:541 then :561): fan_out law, std = sqrt(2/(7*7*64))
    got = dict(zip(names, rows))
    assert abs(got["conv1.weight"][2] / np.sqrt(2.0 / (49 * 64)) - 1.0) < 0.05
    assert abs(ref["conv1.weight"][2] / np.sqrt(2.0 / (49 * 64)) - 1.0) < 0.05


@pytest.mark.parametrize("impl", ["product", "oracle"])
def test_multistage_initialisers_match_reference(impl):
    want = np.load(GOLD)
    if impl == "product":
        from radar_depth_amd.model.multistage_model import ResNet_multistage
    else:
        from oracle.multistage_model import ResNet_multistage
    torch.manual_seed(7)
    names, rows = _moments(ResNet_multistage(18, "upproj", [450, 800], False))
    n_random, n_uniform = _compare(names, rows, want["ms_names"], want["ms_rows"])
    assert n_random == 110 and n_uniform == 4
