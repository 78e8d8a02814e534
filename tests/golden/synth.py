"""Deterministic, platform-independent tensor synthesiser for golden fixtures.

The golden fixtures under tests/golden/ do NOT store model weights (a 3-layer
flow at the real dims is ~17 MB).  Instead both the fixture generator
(gen_golden.py, which loads the values INTO the reference model) and the tests
(which load them into the oracle / the HIP engine) derive every tensor of a
state_dict from (seed, tensor name, shape) with pure 64-bit integer arithmetic
(splitmix64), so the values are bit-identical on every machine and
independent of torch / numpy RNG implementations.
"""
import zlib

import numpy as np
import torch

_MASK = (1 << 64) - 1


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 numpy arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(_MASK)
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(_MASK)
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(_MASK)
    return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (key, seed)."""
    base = (zlib.crc32(key.encode()) * 0x100000001B3 + seed * 0x9E3779B1) & _MASK
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(base)
        bits = _splitmix64(_splitmix64(ctr))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform(key, shape, lo, hi, seed=0):
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (hi - lo) * uniform01(key, n, seed)).reshape(shape)


def normal(key, shape, seed=0):
    """Standard normal via Box-Muller on the deterministic uniforms."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = uniform01(key + "/u1", n, seed)
    u2 = uniform01(key + "/u2", n, seed)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return z.reshape(shape)


def synth_tensor(name: str, ref: torch.Tensor, seed: int = 0) -> torch.Tensor:
    """Value for state_dict entry `name` (shape/dtype taken from `ref`).

    Ranges are chosen so that identity-at-init layers (ActNorm, LinearLU, BN
    running stats) are NOT identities and so that activations stay O(1).
    Returns None for entries that keep their constructor value.
    """
    shape = tuple(ref.shape)
    leaf = name.split(".")[-1]
    if leaf in ("buffer", "loc", "zero", "one", "permutation", "inv_permutation"):
        return None
    if leaf == "scale" and "sample_dist" in name:
        return None
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "initialized":
        return torch.ones(shape, dtype=ref.dtype)
    if leaf == "running_mean":
        v = uniform(name, shape, -0.3, 0.3, seed)
    elif leaf == "running_var":
        v = uniform(name, shape, 0.5, 2.0, seed)
    elif leaf in ("shift", "reshift") and len(shape) == 1 and shape[0] == 1:
        # ExponentialCoupling / ExponentialCombiner scalars (defaults 0)
        v = uniform(name, shape, -0.05, 0.05, seed)
    elif leaf == "scale" and len(shape) == 1 and shape[0] == 1:
        v = uniform(name, shape, 0.10, 0.15, seed)          # default 1/8
    elif leaf == "rescale":
        v = uniform(name, shape, 0.9, 1.1, seed)            # default 1
    elif leaf == "shift":                                    # ActNorm [1,D]
        v = uniform(name, shape, -0.3, 0.3, seed)
    elif leaf == "log_scale":
        v = uniform(name, shape, -0.3, 0.3, seed)
    elif leaf in ("lower_entries", "upper_entries"):
        # D(D-1)/2 entries; keep L·U well conditioned
        d = (1 + int(round((1 + 8 * shape[0]) ** 0.5))) // 2
        a = 0.6 / np.sqrt(d)
        v = uniform(name, shape, -a, a, seed)
    elif leaf == "unconstrained_upper_diag":
        v = uniform(name, shape, 0.0, 1.0, seed)
    elif leaf == "w" and len(shape) == 2:                    # Full/ExponentialCombiner
        d = shape[0]
        v = uniform(name, shape, -1.0, 1.0, seed) * (0.5 / np.sqrt(d)) + np.eye(d) * 0.9
    elif leaf == "weightbank":                               # PAConv [2*C_in, m*C_out]
        a = 1.6 / np.sqrt(shape[0])
        v = uniform(name, shape, -a, a, seed)
    elif leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = 1.6 / np.sqrt(fan_in)
        v = uniform(name, shape, -a, a, seed)
    elif leaf == "weight":                                    # LayerNorm / BatchNorm gamma
        v = uniform(name, shape, 0.6, 1.4, seed)
        if "bn" in name or ".1." in name:                     # BN: some negative gammas
            sgn = np.where(uniform01(name + "/sgn", v.size, seed).reshape(shape) < 0.2, -1.0, 1.0)
            v = v * sgn
    elif leaf == "bias":
        v = uniform(name, shape, -0.1, 0.1, seed)
    else:
        raise KeyError(f"synth_tensor: no rule for state_dict entry {name!r} {shape}")
    return torch.from_numpy(np.ascontiguousarray(v)).to(ref.dtype)


def synth_state_dict(template: dict, seed: int = 0) -> dict:
    """New state_dict with the same keys/shapes/dtypes as `template`."""
    out = {}
    for k, ref in template.items():
        v = synth_tensor(k, ref, seed)
        out[k] = ref.clone() if v is None else v
    return out


def synth_points(key, B, n, seed=0):
    """Synthetic coloured cloud pairs, SURVEY.md §8(d): xyz ~ U(-1,1)^3 then
    centred / max-norm scaled per cloud, rgb ~ U[0,1)."""
    xyz = uniform(key + "/xyz", (B, n, 3), -1.0, 1.0, seed)
    xyz = xyz - xyz.mean(axis=1, keepdims=True)
    xyz = xyz / np.linalg.norm(xyz, axis=-1).max(axis=1)[:, None, None]
    rgb = uniform(key + "/rgb", (B, n, 3), 0.0, 1.0, seed)
    return torch.from_numpy(np.concatenate([xyz, rgb], -1)).float()
