"""Golden vectors for the input-staging row, produced HERE from /root/reference.

The reference's dataset class cannot be constructed in this container (h5py, the nuScenes devkit and the data are absent), but
its transform classes import: the vectors below run the literal statement sequence of get_data / transform_val
(dataset/nuscenes_dataset_torch_new.py:191-195, 417-456, 498-512) with the reference's OWN ``transforms.CenterCrop`` and
``transforms.ToTensor`` objects on seeded frames.      python tests/golden/make_golden_staging.py"""
import os
import sys
import types

import numpy as np
import torch

sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))      # dataset/transforms.py only needs the name
sys.path.insert(0, "/root/reference")
from dataset import transforms  # noqa: E402

to_tensor = transforms.ToTensor()


def reference_val_frame(image, lidar_i16, radar_i16, crop_size_val, max_depth):
    """The op sequence of get_data (:193,:195) and transform_val (:417-456,:498-512) for modality rgbd / sparsifier radar,
    executed with the reference's own CenterCrop and ToTensor objects (the only parts with non-trivial semantics)."""
    crop = transforms.Compose([transforms.CenterCrop(crop_size_val)])
    lidar = np.array(lidar_i16 / 256.).astype(np.float32)            # int16 -> float64 metres -> float32
    radar = np.array(radar_i16 / 256.).astype(np.float32)
    rgb = crop(np.array(image).astype(np.float32)) / 255.            # float32 array / python float
    rgb_t = to_tensor(np.array(rgb).astype(np.float32))             # HWC -> CHW
    lidar_t = to_tensor(np.array(crop(lidar)).astype(np.float32)).unsqueeze(0)
    radar_t = to_tensor(np.array(crop(radar)).astype(np.float32)).unsqueeze(0)
    radar_t[radar_t > max_depth] = 0                                 # tensor-vs-python-scalar comparison, as in :506-507
    return torch.cat((rgb_t, radar_t), dim=0).numpy(), lidar_t.numpy()


def main():
    rng = np.random.RandomState(20260928)
    cases = {}
    # (frames, H0, W0, crop, max_depth): odd margins exercise Python's round-half-to-even in CenterCrop.get_params
    for name, (B, H0, W0, crop, md) in {"a": (2, 13, 21, (8, 12), 80.0), "b": (1, 9, 17, (8, 14), np.inf), "c": (3, 11, 19, (11, 19), 25.5)}.items():
        img = rng.randint(0, 256, size=(B, H0, W0, 3)).astype(np.uint8)
        lidar = (rng.rand(B, H0, W0) * 120.0 * 256 * (rng.rand(B, H0, W0) < 0.3)).astype(np.int16)
        radar = (rng.rand(B, H0, W0) * 120.0 * 256 * (rng.rand(B, H0, W0) < 0.2)).astype(np.int16)
        outs = [reference_val_frame(img[b], lidar[b], radar[b], crop, md) for b in range(B)]
        cases[name + "_image"], cases[name + "_lidar"], cases[name + "_radar"] = img, lidar, radar
        cases[name + "_crop"], cases[name + "_max_depth"] = np.array(crop), np.array(md, dtype=np.float64)
        cases[name + "_inputs"] = np.stack([o[0] for o in outs])
        cases[name + "_labels"] = np.stack([o[1] for o in outs])
    # every byte value through the /255 path
    img = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    z = np.zeros((1, 16, 16), dtype=np.int16)
    i, l = reference_val_frame(img[0], z[0], z[0], (16, 16), np.inf)
    cases["bytes_inputs"] = i[None]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "staging.npz")
    np.savez_compressed(out, **cases)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
