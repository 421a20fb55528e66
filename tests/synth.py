"""Deterministic synthetic inputs shared by the golden generator, the tests and bench.py.

Integer-hash based (no RNG library state) so the *same* tables/weights can be re-created
bit-exactly in any process, on any numpy version, here and on the GPU box.
"""
from __future__ import annotations

import numpy as np


def _u01(idx: np.ndarray, seed: int) -> np.ndarray:
    """uniform [0,1) from a 32-bit integer mix (murmur3 finaliser)."""
    h = (idx.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return (h.astype(np.float64) / 4294967296.0)


def uniform(shape, lo: float, hi: float, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * _u01(np.arange(n, dtype=np.uint64), seed)).astype(np.float32).reshape(shape)


def normal(shape, seed: int) -> np.ndarray:
    n = int(np.prod(shape))
    u1 = _u01(np.arange(n, dtype=np.uint64), seed) + 1e-12
    u2 = _u01(np.arange(n, dtype=np.uint64), seed + 7919)
    return (np.sqrt(-2 * np.log(u1)) * np.cos(2 * np.pi * u2)).astype(np.float32).reshape(shape)


def hash_table(n_rows: int, n_feat: int, seed: int, scale: float = 1e-3) -> np.ndarray:
    """U(-scale, scale) like encodings.py:382-384 (hash_init_scale)."""
    return uniform((n_rows, n_feat), -scale, scale, seed)


def linear(out_dim: int, in_dim: int, seed: int, bias: bool = True):
    """nn.Linear default init range U(-1/sqrt(in), 1/sqrt(in))."""
    k = 1.0 / np.sqrt(in_dim)
    w = uniform((out_dim, in_dim), -k, k, seed)
    b = uniform((out_dim,), -k, k, seed + 1) if bias else None
    return w, b


def rays(n_rays: int, seed: int, lidar_frac: float = 0.0):
    """SURVEY.md §8(d) synthetic rays: origins N(0,5^2) m, unit directions, camera/lidar pixel_area."""
    o = normal((n_rays, 3), seed) * 5.0
    d = normal((n_rays, 3), seed + 1)
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    area = np.full((n_rays,), 2.7e-7 * 9, np.float32)
    n_l = int(n_rays * lidar_frac)
    if n_l:
        area[n_rays - n_l:] = 4.5e-6
    t = uniform((n_rays,), 0.0, 8.0, seed + 2)
    return o.astype(np.float32), d.astype(np.float32), area, t
