"""Counter-based synthetic data and procedural weight fill.

Everything here is a pure function of (name/seed, shape): no torch RNG streams, no
files.  The same functions fill the imported reference (when the golden fixtures are
generated), the CPU oracle and the HIP-backed modules, so parity tests never have to
ship a 58.8 MB state_dict and the bench can build its batches on any
rank (seed = 1234 + 1000*rank + iter).

Input recipe (reference dataset contract
dataset/nuscenes_dataset_torch_new.py:304,332-333,373-375 and
dataset/dense_to_sparse.py:79):
  inputs [B,4,H,W] fp32: ch0-2 RGB ~ U[0,1); ch3 radar depth (metres) = Bernoulli(100/(450*800)) * U(0,80]
  target [B,1,H,W] fp32: lidar depth = Bernoulli(3000/(450*800)) * U(0,80], 0 = invalid
"""
import zlib

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def uniform01(n, seed, stream=0):
    """n doubles in [0,1) from counter-based hashing of (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        key = _mix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64))[0]
        h = _mix64(idx ^ key)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal01(n, seed, stream=0):
    """n standard normals (Box-Muller on two hashed uniform streams)."""
    u1 = uniform01(n, seed, 2 * stream + 101)
    u2 = uniform01(n, seed, 2 * stream + 102)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return r * np.cos(2.0 * np.pi * u2)


def name_seed(name):
    return zlib.crc32(name.encode("utf-8")) & 0x7FFFFFFF


def fill_tensor_by_name(name, tensor):
    """Deterministic values for a state_dict entry from its key and shape.

    conv weights ([O,I,kh,kw]): N(0, sqrt(2/(kh*kw*O)))  (the scale of models.py:33-34)
    BN weight: 1 + 0.2*(u-0.5); BN bias: 0.2*(u-0.5); running_mean: 0.1*n; running_var: 0.8+0.4u
    scalar parameters (w_stage1 / w_stage2): 1 + 0.1*(u-0.5); num_batches_tracked: 0
    """
    shape = tuple(tensor.shape)
    n = int(np.prod(shape)) if len(shape) else 1
    seed = name_seed(name)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        vals = np.zeros(n)
    elif len(shape) == 4:
        o, _, kh, kw = shape
        vals = normal01(n, seed) * np.sqrt(2.0 / (kh * kw * o))
    elif leaf == "running_mean":
        vals = 0.1 * normal01(n, seed)
    elif leaf == "running_var":
        vals = 0.8 + 0.4 * uniform01(n, seed)
    elif leaf == "weight":
        vals = 1.0 + 0.2 * (uniform01(n, seed) - 0.5)
    elif leaf == "bias":
        vals = 0.2 * (uniform01(n, seed) - 0.5)
    else:  # w_stage1, w_stage2 and any other scalar
        vals = 1.0 + 0.1 * (uniform01(n, seed) - 0.5)
    arr = torch.from_numpy(np.asarray(vals).reshape(shape if len(shape) else ()))
    with torch.no_grad():
        tensor.copy_(arr.to(tensor.dtype))
    return tensor


def procedural_fill_(module):
    """Fill every entry of module.state_dict() in place by key name (see fill_tensor_by_name)."""
    for key, value in module.state_dict().items():
        fill_tensor_by_name(key, value)
    return module


def make_batch(batch, height, width, seed, device="cpu", radar_points=100.0, lidar_points=3000.0,
               ref_pixels=450 * 800, max_depth=80.0):
    """Synthetic (inputs [B,4,H,W], target [B,1,H,W]) fp32 following the recipe in the module docstring.

    The point densities are per reference-sized frame (450x800), so smaller test geometries keep
    the same sparsity.
    """
    hw = height * width
    n_in = batch * 4 * hw
    u = uniform01(n_in, seed, 1).reshape(batch, 4, height, width)
    inputs = np.empty((batch, 4, height, width), dtype=np.float32)
    inputs[:, :3] = u[:, :3]
    p_radar = radar_points / ref_pixels
    hit = u[:, 3] < p_radar
    depth = (1.0 - uniform01(batch * hw, seed, 2).reshape(batch, height, width)) * max_depth
    inputs[:, 3] = np.where(hit, depth, 0.0)
    p_lidar = lidar_points / ref_pixels
    ut = uniform01(batch * hw, seed, 3).reshape(batch, 1, height, width)
    dt = (1.0 - uniform01(batch * hw, seed, 4).reshape(batch, 1, height, width)) * max_depth
    target = np.where(ut < p_lidar, dt, 0.0).astype(np.float32)
    return torch.from_numpy(inputs).to(device), torch.from_numpy(target).to(device)
