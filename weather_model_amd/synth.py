"""Counter-based synthetic data (host side, numpy).

One 64-bit hash of (seed, element index) per element -- the same function the HIP generator in
csrc/synth.hip evaluates on the device -- so a tensor's contents depend only on (seed, index),
never on how many GPUs or chunks produced it (SURVEY.md 7 "Config 5 does not fit", 8e).

uniform01: 24 random bits -> exact float32 in [0, 1).
normal   : Box-Muller on two such uniforms (device and host agree to float32 rounding only,
           so bit-exact comparisons must use uniform data or download the device tensor).
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def hash64(seed: int, idx: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser of (idx + (seed+1) * golden)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((int(seed) + 1) & 0xFFFFFFFF) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """float32 U[0,1) for element indices offset .. offset+count-1."""
    idx = np.arange(offset, offset + count, dtype=np.uint64)
    z = hash64(seed, idx)
    return ((z >> np.uint64(40)).astype(np.float32)) * np.float32(2.0 ** -24)


def uniform_paired01(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """float32 U[0,1), two 24-bit draws per hash (csrc/synth.hip, dist 2): elements 2k and 2k+1 take
    bits 63..40 and 39..16 of hash64(seed, k).  A different stream from uniform01."""
    idx = np.arange(offset, offset + count, dtype=np.uint64)
    z = hash64(seed, idx >> np.uint64(1))
    hi = z >> np.uint64(40)
    lo = (z >> np.uint64(16)) & np.uint64(0xFFFFFF)
    bits = np.where((idx & np.uint64(1)).astype(bool), lo, hi)
    return bits.astype(np.float32) * np.float32(2.0 ** -24)


def uniform32_01(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """float32 U[0,1) from one 32-bit hash per element (csrc/synth.hip, dist 3): the lowbias32
    finaliser of a key that folds the 64-bit element index with odd multipliers.  A different
    stream from uniform01 / uniform_paired01."""
    idx = np.arange(offset, offset + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = (idx & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        k = k + np.uint32(0x9E3779B9) * (idx >> np.uint64(32)).astype(np.uint32)
        k = k + np.uint32(0x85EBCA6B) * np.uint32(int(seed) & 0xFFFFFFFF) + np.uint32(0x165667B1)
        k ^= k >> np.uint32(16)
        k *= np.uint32(0x7FEB352D)
        k ^= k >> np.uint32(15)
        k *= np.uint32(0x846CA68B)
        k ^= k >> np.uint32(16)
    return (k >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def uniform(seed: int, shape, lo=0.0, hi=1.0, offset: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    u = uniform01(seed, n, offset)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32).reshape(shape)


def normal(seed: int, shape, mean=0.0, std=1.0, offset: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    z = hash64(seed, idx)
    u1 = ((z >> np.uint64(40)).astype(np.float64) + 1.0) * 2.0 ** -24     # (0,1]
    u2 = ((z >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.float64) * 2.0 ** -24
    g = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return (mean + std * g).astype(np.float32).reshape(shape)


def regression_problem(n: int, F: int, P: int = 1, seed: int = 1, dist: str = "uniform",
                       noise: float = 0.01, row_offset: int = 0):
    """Rows of the synthetic least-squares problem of SURVEY.md 8d config 2/5:
    X ~ U(0,1) (mimics normalize, multipleRegression.py:190-192) or N(0,1) (standardize,
    :195-197); beta* ~ N(0,1) (seed+1000), intercept 0.5; Y = X beta* + 0.5 + N(0, noise^2).
    Row r of X uses element indices r*F .. r*F+F-1, so any row range can be produced alone."""
    beta = normal(seed + 1000, (F, P)).astype(np.float64)
    if dist == "uniform":
        X = uniform(seed, (n, F), offset=row_offset * F)
    elif dist == "uniform_paired":
        X = uniform_paired01(seed, n * F, offset=row_offset * F).reshape(n, F)
    elif dist == "uniform32":
        X = uniform32_01(seed, n * F, offset=row_offset * F).reshape(n, F)
    else:
        X = normal(seed, (n, F), offset=row_offset * F)
    eps = normal(seed + 2000, (n, P), std=noise, offset=row_offset * P)
    Y = (X.astype(np.float64) @ beta + 0.5).astype(np.float32) + eps
    return X, Y.astype(np.float32), beta
