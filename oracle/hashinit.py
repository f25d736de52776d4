"""Bit-reproducible weight / input generator (TEST INFRASTRUCTURE, see oracle/__init__.py).

Every tensor is a pure function of (name, shape, scale): element i of tensor ``name`` is
``scale * u`` with ``u`` uniform in [-1, 1) derived from splitmix64(fnv1a64(name) + i).
The real reference (in the build container), the CPU restatement and the HIP path all
fill their parameters through this generator, so golden fixtures never have to carry
weights and the GPU box can regenerate them without the reference.
"""
from __future__ import annotations

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(name: str, shape, scale: float = 1.0) -> np.ndarray:
    """float32 array of ``shape`` with values scale * U[-1, 1)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(fnv1a64(name))) & _MASK
    bits = splitmix64(idx) >> np.uint64(40)  # 24 random bits
    u = bits.astype(np.float64) * (2.0 / float(1 << 24)) - 1.0
    return (u * scale).astype(np.float32).reshape(shape)


def param(name: str, shape) -> np.ndarray:
    """Parameter fill used for parity runs.

    LayerNorm weights ~ 1 + 0.1 u, biases ~ 0.05 u, everything else 0.06 u
    (std ≈ 0.035, the same order as the reference's N(0, 0.02) initialisers, large
    enough that every code path — bias tables, padding rows, dead parameters —
    contributes visibly to the outputs).
    """
    lname = name.lower()
    is_ln = ("layernorm" in lname or "layer_norm" in lname or ".layernorm_" in lname)
    if is_ln and name.endswith(".weight"):
        return (1.0 + uniform(name, shape, 0.1)).astype(np.float32)
    if name.endswith(".bias"):
        return uniform(name, shape, 0.05)
    return uniform(name, shape, 0.06)
