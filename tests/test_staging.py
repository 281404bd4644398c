"""Input staging: oracle pinned to the reference's own transform classes (tests/golden/staging.npz);
the HIP kernel bit-exact against the oracle and the golden vectors, through the C ABI."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "staging.npz"))


def _case(name):
    return (G[name + "_image"], G[name + "_lidar"], G[name + "_radar"], tuple(int(v) for v in G[name + "_crop"]),
            float(G[name + "_max_depth"]), G[name + "_inputs"], G[name + "_labels"])


@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_oracle_matches_reference_vectors(name):
    from oracle.staging import stage_val_batch
    img, lidar, radar, crop, md, want_in, want_lb = _case(name)
    got_in, got_lb = stage_val_batch(img, lidar, radar, crop, md if np.isfinite(md) else -1.0)
    assert got_in.dtype == np.float32 and got_lb.dtype == np.float32
    assert np.array_equal(got_in, want_in) and np.array_equal(got_lb, want_lb)      # integer/byte work: bit-exact


def test_oracle_every_byte_value():
    from oracle.staging import stage_val_batch
    img = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    z = np.zeros((1, 16, 16), dtype=np.int16)
    got, _ = stage_val_batch(img, z, z, (16, 16), -1.0)
    assert np.array_equal(got, G["bytes_inputs"])


def test_center_crop_params_round_half_to_even():
    from radar_depth_amd.dataset.staging import center_crop_params
    assert center_crop_params(13, 21, (8, 12)) == (2, 4, 8, 12)          # 2.5 -> 2, 4.5 -> 4 (Python round)
    assert center_crop_params(9, 17, (8, 14)) == (0, 2, 8, 14)           # 0.5 -> 0, 1.5 -> 2
    assert center_crop_params(450, 800, (450, 800)) == (0, 0, 450, 800)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b", "c"])
def test_hip_staging_matches_golden(name):
    from radar_depth_amd.dataset import stage_val_batch
    img, lidar, radar, crop, md, want_in, want_lb = _case(name)
    x, y = stage_val_batch(torch.from_numpy(img).cuda(), torch.from_numpy(lidar).cuda(), torch.from_numpy(radar).cuda(), crop,
                           md if np.isfinite(md) else -1.0)
    assert np.array_equal(x.cpu().numpy(), want_in) and np.array_equal(y.cpu().numpy(), want_lb)


@pytest.mark.gpu
def test_hip_staging_every_byte_value():
    from radar_depth_amd.dataset import stage_val_batch
    img = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1).repeat(1, 1, 1, 3).cuda()
    z = torch.zeros(1, 16, 16, dtype=torch.int16).cuda()
    x, _ = stage_val_batch(img, z, z, (16, 16), -1.0)
    assert np.array_equal(x.cpu().numpy(), G["bytes_inputs"])


@pytest.mark.gpu
@pytest.mark.parametrize("B,H0,W0,crop,md", [(16, 450, 800, (450, 800), 80.0), (4, 455, 803, (450, 800), -1.0), (2, 900, 1600, (450, 800), 50.0),
                                             (3, 17, 23, (9, 10), 30.0)])
def test_hip_staging_full_size_vs_oracle(B, H0, W0, crop, md):
    """BASELINE.json's batch geometry (and ragged widths that take the scalar store path): bit-exact vs the oracle, plus the
    size-independent properties: labels*256 round-trips the int16 input, masked radar never exceeds max_depth."""
    from oracle.staging import stage_val_batch as oracle_stage
    from radar_depth_amd.dataset import center_crop_params, stage_val_batch
    rng = np.random.RandomState(B * 1000 + H0)
    img = rng.randint(0, 256, size=(B, H0, W0, 3)).astype(np.uint8)
    lidar = (rng.rand(B, H0, W0) * 100 * 256 * (rng.rand(B, H0, W0) < 0.3)).astype(np.int16)
    radar = (rng.rand(B, H0, W0) * 100 * 256 * (rng.rand(B, H0, W0) < 0.2)).astype(np.int16)
    x, y = stage_val_batch(torch.from_numpy(img).cuda(), torch.from_numpy(lidar).cuda(), torch.from_numpy(radar).cuda(), crop, md)
    want_x, want_y = oracle_stage(img, lidar, radar, crop, md)
    x, y = x.cpu().numpy(), y.cpu().numpy()
    assert np.array_equal(x, want_x) and np.array_equal(y, want_y)
    i0, j0, th, tw = center_crop_params(H0, W0, crop)
    assert np.array_equal((y[:, 0] * 256).astype(np.int16), lidar[:, i0:i0 + th, j0:j0 + tw])
    if md >= 0:
        assert x[:, 3].max() <= md


@pytest.mark.gpu
def test_hip_staging_rejects_bad_crop():
    from radar_depth_amd._lib import RadarDepthHipError
    from radar_depth_amd.dataset import stage_val_batch
    img = torch.zeros(1, 8, 8, 3, dtype=torch.uint8).cuda()
    z = torch.zeros(1, 8, 8, dtype=torch.int16).cuda()
    with pytest.raises(RadarDepthHipError):
        stage_val_batch(img, z, z, (9, 8))
