"""Pins the CPU oracle to the real reference: the oracle re-runs the exact case functions that
tests/golden/make_golden.py ran on the imported reference and must reproduce the committed vectors.
Same torch ops on both sides -> tolerance 1e-6 relative."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import criteria, models, multistage_model

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def check_all(got, want, tol, skip=()):
    for k in want.files:
        if k in skip or want[k].dtype.kind in "US":
            continue
        assert k in got, k
        if want[k].dtype.kind == "b":
            assert (got[k] == want[k]).all(), k
        else:
            assert rel(got[k], want[k]) <= tol, (k, rel(got[k], want[k]))


def test_latefusion_small(golden_dir):
    want = np.load(os.path.join(golden_dir, "latefusion_small.npz"))
    got = mg.latefusion_case(models.ResNet_latefusion, criteria.MaskedL1Loss, 2, 97, 161, 4321, 1, True)
    assert list(got["param_names"]) == list(want["param_names"])
    assert list(got["stat_names"]) == [s for s in want["stat_names"] if "unpool" not in s] or \
        set(got["stat_names"]) >= set(s for s in want["stat_names"])
    gs = dict(zip(got["stat_names"], got["stat_values"]))
    for name, val in zip(want["stat_names"], want["stat_values"]):
        assert rel(gs[name], val) <= 1e-5, name
    check_all(got, want, 2e-6, skip=("stat_values",))


def test_latefusion_full(golden_dir):
    want = np.load(os.path.join(golden_dir, "latefusion_full.npz"))
    got = mg.latefusion_case(models.ResNet_latefusion, criteria.MaskedL1Loss, 2, 450, 800, 1234, 8, False)
    gs = dict(zip(got["stat_names"], got["stat_values"]))
    for name, val in zip(want["stat_names"], want["stat_values"]):
        assert rel(gs[name], val) <= 1e-5, name
    check_all(got, want, 5e-6, skip=("stat_values",))


def test_multistage_small(golden_dir):
    want = np.load(os.path.join(golden_dir, "multistage_small.npz"))
    got = mg.multistage_case(multistage_model.ResNet_multistage, criteria.MaskedL1Loss, criteria.SmoothnessLoss,
                             2, 97, 161, 777, 1)
    assert list(got["param_names"]) == list(want["param_names"])
    assert list(want["param_names"][:2]) == ["w_stage1", "w_stage2"]
    assert (got["out/mask"] != want["out/mask"]).mean() == 0
    assert 0.0 < float(want["mask_density"][0]) < 1.0
    assert float(want["coupling_norm"][0]) > 0
    check_all(got, want, 5e-6)


def test_units(golden_dir):
    want = np.load(os.path.join(golden_dir, "units.npz"))
    got = mg.unit_cases(multistage_model, criteria)
    check_all(got, want, 1e-6)
    assert bool(want["l1/empty_isnan"][0])


def test_upproj_module(golden_dir):
    want = np.load(os.path.join(golden_dir, "upproj_module.npz"))
    got = mg.upproj_case(models)
    check_all(got, want, 2e-6)


def test_basic_block(golden_dir):
    want = np.load(os.path.join(golden_dir, "basic_block.npz"))
    got = mg.block_case(models)
    check_all(got, want, 2e-6)


def test_multistage_full(golden_dir):
    """BASELINE config 4's geometry (450x800): the oracle reproduces the reference's multistage outputs, losses and gradients."""
    want = np.load(os.path.join(golden_dir, "multistage_full.npz"))
    got = mg.multistage_case(multistage_model.ResNet_multistage, criteria.MaskedL1Loss, criteria.SmoothnessLoss,
                             2, 450, 800, 4242, 8, dense_small=False)
    assert (got["out/mask"] != want["out/mask"]).mean() == 0
    check_all(got, want, 1e-5)


def test_state_dict_contract():
    m = models.ResNet_latefusion(18, "upproj", [450, 800], 4, False)
    sd = m.state_dict()
    assert len(sd) == 325 and sum(p.numel() for p in m.parameters()) == 14710752
    assert not any("unpool" in k for k in sd)
    assert sd["conv_fusion.weight"].shape == (512, 640, 1, 1)
    ms = multistage_model.ResNet_multistage(18, "upproj", [450, 800], False)
    assert ms.stage2.conv1_depth.weight.shape == (16, 2, 7, 7)
    with pytest.raises(RuntimeError):
        models.ResNet_latefusion(19, "upproj", [450, 800])
    with pytest.raises(AssertionError):
        models.ResNet_latefusion(18, "upproj", [450, 800], in_channels=3, pretrained=False)
    with pytest.raises(AssertionError):
        models.choose_decoder("nope", 256)
    with pytest.raises(ValueError):
        multistage_model.ResNet_multistage(18, "upproj", [450, 800], True)


def test_metrics_oracle(golden_dir):
    from oracle import metrics
    want = np.load(os.path.join(golden_dir, "metrics.npz"))
    got = metrics.evaluate(torch.tensor(want["out"]), torch.tensor(want["target"]))
    assert rel(got, want["r1"]) < 1e-6
