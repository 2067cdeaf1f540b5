"""Deterministic, platform-independent synthetic data (test infrastructure).

Values come from integer hashing (splitmix64 finaliser) of the flat element
index, so the same arrays are reproduced bit-for-bit in the build container
(where the golden fixtures are generated from the reference) and on the GPU box
(where the reference does not exist).  No RNG-implementation dependence.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(shape, salt, lo=-1.0, hi=1.0):
    """float32 array of `shape`, uniform in [lo, hi), keyed by integer `salt`."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(salt) * np.uint64(0x1000003D)) & _M64
        h = _mix(_mix(idx) ^ np.uint64(salt))
    u = (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)  # [0,1)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(shape, salt):
    """Zero-mean, unit-variance, bell-shaped (sum of 4 uniforms) float32 array."""
    acc = np.zeros(shape, dtype=np.float64)
    for k in range(4):
        acc += uniform(shape, salt * 4 + k + 1000003).astype(np.float64)
    return (acc * np.sqrt(3.0 / 4.0)).astype(np.float32)


def t_uniform(shape, salt, lo=-1.0, hi=1.0):
    return torch.from_numpy(uniform(tuple(shape), salt, lo, hi))


def t_normalish(shape, salt):
    return torch.from_numpy(normalish(tuple(shape), salt))


def stereo_features(B, C, H, W, salt, max_shift=6):
    """A (left, right) feature pair where right is left shifted by a smooth,
    signed, per-row integer disparity plus small noise, so that correlation
    volumes have a real peak (used for hot-segment runs)."""
    left = normalish((B, C, H, W), salt)
    right = np.zeros_like(left)
    ys = np.arange(H)
    shift = np.rint(max_shift * np.sin(2.0 * np.pi * ys / max(H, 1) + 0.3 * salt)).astype(np.int64)
    for y in range(H):
        # left[x] matches right[x - d]  =>  right[x] = left[x + d]
        right[:, :, y, :] = np.roll(left[:, :, y, :], -int(shift[y]), axis=-1)
    right = right + 0.05 * normalish((B, C, H, W), salt + 77)
    return torch.from_numpy(left), torch.from_numpy(right.astype(np.float32))


def distinct_sorted_candidates(B, k, H, W, m, salt):
    """[B,k,H,W] float32: per pixel, k distinct integers in [-m, m), ascending
    (the form `disparity_sample_topk` has at models/SemStereo.py:305)."""
    D = 2 * m
    assert k <= D
    keys = uniform((B, D, H, W), salt)
    order = np.argsort(keys, axis=1, kind="stable")[:, :k]
    cand = np.sort(order, axis=1).astype(np.float32) - float(m)
    return torch.from_numpy(cand)
