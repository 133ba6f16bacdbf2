"""Synthetic inputs for the parity tests and bench.py (no datasets exist for this path).

Scalars are produced directly as Montgomery limbs: a splitmix64 stream fills 4 limbs per scalar and the top limb
is reduced below q's top limb, so every output is a valid fully-reduced BlsScalar image (the Montgomery map is a
bijection of the field, so uniform limbs are a uniform field element up to the excluded top sliver)."""
from __future__ import annotations

import numpy as np

Q = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % Q
Q_TOP = Q >> 192
SEED = 0x706C6F6E6B5F6761  # "plonk_ga"


def splitmix64(n: int, seed: int = SEED) -> np.ndarray:
    """n consecutive outputs of splitmix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        z = (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_scalars(n: int, seed: int = SEED) -> np.ndarray:
    """uint64[n,4] Montgomery limbs of n pseudo-random field elements."""
    a = splitmix64(4 * n, seed).reshape(n, 4).copy()
    a[:, 3] %= np.uint64(Q_TOP)
    return a


def mont(x: int) -> list[int]:
    m = (x % Q) * R % Q
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def scalars_from_ints(xs) -> np.ndarray:
    return np.array([mont(int(x)) for x in xs], dtype=np.uint64).reshape(-1, 4)


def to_int(limbs) -> int:
    """Montgomery limbs -> canonical integer"""
    m = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return m * pow(R, -1, Q) % Q


def uniform_below(n: int, bound: int, seed: int = SEED) -> np.ndarray:
    """n scalars whose canonical values are pseudo-random in [0, bound) (small n: Python ints)."""
    d = splitmix64(5 * n, seed).reshape(n, 5)
    vals = [sum(int(d[i, k]) << (64 * k) for k in range(5)) % bound for i in range(n)]
    return scalars_from_ints(vals)
