"""No-GPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/radar_depth_hip.h; argument validation returns error codes (no compute); the plugin surface
(parse_command / create_model / constructors) mirrors the reference's names, defaults and error behaviour."""
import ctypes as C
import os
import re
import types

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from radar_depth_amd.build import build
    build(verbose=False)
    from radar_depth_amd._lib import lib
    return lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(REPO, "include", "radar_depth_hip.h")).read()
    names = sorted(set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 45
    assert [n for n in names if not hasattr(L, n)] == []
    assert L.rd_abi_version() == 1


def test_argument_validation_without_gpu(L):
    from radar_depth_amd import convdesc as cd
    d = cd.conv_fwd(2, 9, 7, 24, 8, 3, 1, 1)            # Cin not a multiple of 16
    assert L.rd_gconv_stat_tiles(C.byref(d)) < 0
    assert b"Cin" in L.rd_last_error()
    assert L.rd_gconv(C.byref(cd.conv_fwd(2, 9, 7, 32, 8, 3, 1, 1)), None, None, None, None, 0, None, None) == -1
    assert L.rd_sgd_step(None, None, None, C.c_int64(4), C.c_float(0.1), C.c_float(0.9), C.c_float(0.0), C.c_float(1.0), 0, None) == -1
    d = cd.upproj_fwd(2, 15, 25, 256, 256)
    assert L.rd_gconv_stat_tiles(C.byref(d)) > 0 and L.rd_wgrad_workspace_floats(C.byref(d)) > 0


def test_plugin_surface():
    from radar_depth_amd import main as hmain, utils
    from radar_depth_amd.model import models, multistage_model
    a = utils.parse_command(["-a", "resnet18_latefusion", "-d", "upproj", "-m", "rgbd", "--data", "nuscenes", "--no-pretrain", "-b", "16"])
    assert (a.arch, a.decoder, a.modality, a.batch_size, a.pretrained, a.lr, a.momentum, a.weight_decay) == \
        ("resnet18_latefusion", "upproj", "rgbd", 16, False, 0.01, 0.9, 1e-4)
    assert models.Decoder.names == ["deconv2", "deconv3", "upconv", "upproj"]
    m = hmain.create_model(a, [450, 800])
    assert isinstance(m, models.ResNet_latefusion) and len(m.state_dict()) == 325
    a.arch = "resnet18_multistage_uncertainty_fixs"
    m2, lw = hmain.create_model(a, [450, 800])
    assert isinstance(m2, multistage_model.ResNet_multistage) and lw["w_smooth"] == 0.1
    assert [n for n, _ in m2.named_parameters()][:2] == ["w_stage1", "w_stage2"] and len(m2.state_dict()) == 652
    with pytest.raises(ValueError, match="Unknown model"):
        hmain.create_model(types.SimpleNamespace(arch="nope", decoder="upproj", modality="rgbd", pretrained=False), [450, 800])
    with pytest.raises(RuntimeError):
        models.ResNet_latefusion(19, "upproj", [450, 800])
    with pytest.raises(AssertionError):
        models.ResNet_latefusion(18, "upproj", [450, 800], in_channels=3, pretrained=False)
    with pytest.raises(AssertionError):
        models.choose_decoder("deconv9x", 256)
    with pytest.raises(ValueError, match="pretrained latefusion"):
        multistage_model.ResNet_multistage(18, "upproj", [450, 800], True)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 4, 450, 800))
    opt = torch.optim.SGD(m.parameters(), 0.01)
    assert utils.adjust_learning_rate(opt, 7, 0.01) == pytest.approx(0.001) and opt.param_groups[0]["lr"] == pytest.approx(0.001)


def test_checkpoint_wire_format(tmp_path):
    """the reference's .pth.tar payload round-trips between the HIP-backed modules and the
    reference-shaped (oracle) modules: identical keys, shapes and values, multistage shape filtering included."""
    from oracle.models import ResNet_latefusion as ORef
    from oracle.multistage_model import ResNet_multistage as OMulti
    from radar_depth_amd import utils
    from radar_depth_amd.model.models import ResNet_latefusion
    from radar_depth_amd.model.multistage_model import ResNet_multistage
    from radar_depth_amd.synthetic import procedural_fill_
    src = ORef(18, "upproj", [97, 161], 4, False)
    procedural_fill_(src)
    opt = torch.optim.SGD(src.parameters(), 0.01, momentum=0.9, weight_decay=1e-4)
    path = utils.save_checkpoint({"args": None, "epoch": 3, "arch": "resnet18_latefusion", "model_state_dict": src.state_dict(),
                                  "best_result": None, "optimizer_state_dict": opt.state_dict()}, True, 3, str(tmp_path))
    assert os.path.exists(os.path.join(str(tmp_path), "model_best.pth.tar"))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    m = ResNet_latefusion(18, "upproj", [97, 161], 4, False)
    missing, unexpected = m.load_state_dict(ck["model_state_dict"], strict=False)     # main.py:208,251 use strict=False
    assert not missing and not unexpected
    for (k, a), (k2, b) in zip(m.state_dict().items(), src.state_dict().items()):
        assert k == k2 and torch.equal(a, b)
    # multistage: stage 1 loads the late-fusion checkpoint directly, stage 2 drops the shape-mismatched depth stem
    ms = ResNet_multistage(18, "upproj", [97, 161], False)
    ms.stage1.load_state_dict(ck["model_state_dict"])
    filt = ms.filter_state_dict(dict(ck["model_state_dict"]), ms.stage2.state_dict())
    assert "conv1_depth.weight" not in filt and len(filt) == 324
    ms.stage2.load_state_dict(filt, strict=False)
    assert torch.equal(ms.stage2.conv3.weight, src.conv3.weight)
    # and back: a HIP-module checkpoint loads into the reference-shaped module
    om = OMulti(18, "upproj", [97, 161], False)
    om.load_state_dict(ms.state_dict())
