"""Golden vectors for the input-staging row, produced HERE from /root/reference.

The vectors come from the reference's OWN ``nuscenes_dataset_torch.transform_val`` (dataset/nuscenes_dataset_torch_new.py:415-530),
called unbound on seeded frames with a namespace standing in for ``self`` (it only reads transform_mode / sparsifier / modality /
max_depth / t_cfg.crop_size_*).  The dataset class itself cannot be constructed in this container -- h5py, attrdict, matplotlib,
the nuScenes devkit and the data are absent -- so the module is imported with EMPTY stand-in modules for those packages (none of
them is touched by transform_val, which is numpy / torch and the reference's own ``transforms`` classes), plus the two aliases the
reference's 2019-era code expects (``np.int``, ``collections.Iterable``).  The only statements restated here are get_data's two
decompression lines (:193,:195: ``int16 / 256.``), because get_data reads its input through h5py.  ``reference_val_frame`` -- the
statement sequence the first version of this script restated -- is kept as a cross-check: both must agree bit for bit.
    python tests/golden/make_golden_staging.py"""
import collections
import collections.abc
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch


class _Empty(types.ModuleType):
    """Stand-in for a package the image lacks: any attribute is another empty stand-in, calling it returns one."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        d = _Empty(self.__name__ + "." + name)
        setattr(self, name, d)
        return d

    def __call__(self, *a, **k):
        return _Empty("call")


class _MissingPackages(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    NAMES = ("h5py", "nuscenes", "matplotlib", "pyquaternion", "cv2", "PIL", "torchvision", "skimage", "ipdb", "attrdict")

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.NAMES:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _Empty(spec.name)

    def exec_module(self, module):
        module.__path__ = []


sys.meta_path.insert(0, _MissingPackages())
collections.Iterable = collections.abc.Iterable       # removed in Python 3.10; dataset/transforms.py:313 uses it
np.int = int                                          # removed in numpy 1.24; used on the (unused here) index-map path
sys.path.insert(0, "/root/reference")
from dataset import transforms  # noqa: E402
ref_dataset = importlib.import_module("dataset.nuscenes_dataset_torch_new")


def reference_transform_val(image, lidar_i16, radar_i16, crop_size_val, max_depth):
    """get_data's decompression (:193,:195) + the reference's own transform_val for modality rgbd / sparsifier radar."""
    self = types.SimpleNamespace(transform_mode="sparse-to-dense", sparsifier="radar", modality="rgbd", max_depth=max_depth,
                                 t_cfg=types.SimpleNamespace(crop_size_train=tuple(crop_size_val), crop_size_val=tuple(crop_size_val)))
    data = {"image": image, "lidar_depth": lidar_i16 / 256., "radar_depth": radar_i16 / 256.}
    out = ref_dataset.nuscenes_dataset_torch.transform_val(self, data)
    return out["inputs"].numpy(), out["labels"].numpy()

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
        outs = [reference_transform_val(img[b], lidar[b], radar[b], crop, md) for b in range(B)]
        for b in range(B):          # the restated sequence agrees with the reference's method bit for bit
            chk = reference_val_frame(img[b], lidar[b], radar[b], crop, md)
            assert np.array_equal(outs[b][0], chk[0]) and np.array_equal(outs[b][1], chk[1])
        cases[name + "_image"], cases[name + "_lidar"], cases[name + "_radar"] = img, lidar, radar
        cases[name + "_crop"], cases[name + "_max_depth"] = np.array(crop), np.array(md, dtype=np.float64)
        cases[name + "_inputs"] = np.stack([o[0] for o in outs])
        cases[name + "_labels"] = np.stack([o[1] for o in outs])
    # every byte value through the /255 path
    img = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    z = np.zeros((1, 16, 16), dtype=np.int16)
    i, l = reference_transform_val(img[0], z[0], z[0], (16, 16), np.inf)
    cases["bytes_inputs"] = i[None]
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "staging.npz")
    np.savez_compressed(out, **cases)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
