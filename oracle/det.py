"""Deterministic, seed-free data recipes shared by the capture script, the oracle tests and bench.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

torch.manual_seed() streams are not guaranteed portable across torch builds, and the
golden fixtures must stay small, so weights / noise for the larger parity cases are
generated from a pure-integer hash (splitmix64 finaliser) of (tag, element index).
Integer arithmetic is exact everywhere; the float64 -> float32 casts are exact for the
uniform recipe.  The normal recipe goes through float64 log/cos (libm): any 1-ulp float64
difference vanishes in the float32 cast except on measure-zero ties, which a tolerance-based
test does not see.
"""
from __future__ import annotations

import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def _key(tag: str) -> np.uint64:
    b = tag.encode("utf-8")
    lo = zlib.crc32(b) & 0xFFFFFFFF
    hi = zlib.crc32(b[::-1] + b"#") & 0xFFFFFFFF
    return np.uint64((hi << 32) | lo)


def det_u01(tag: str, n: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in the open interval (0, 1)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = _splitmix64(np.array([_key(tag) ^ np.uint64(stream * 0x632BE59BD9B4E019 & 0xFFFFFFFFFFFFFFFF)], dtype=np.uint64))[0]
        h = _splitmix64(idx * np.uint64(0xD1342543DE82EF95) + k)
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def det_uniform(tag: str, shape, scale: float = 1.0) -> np.ndarray:
    """float32 uniform in [-scale, scale)."""
    n = int(np.prod(shape))
    u = det_u01(tag, n)
    return ((2.0 * u - 1.0) * scale).astype(np.float32).reshape(shape)


def det_normal(tag: str, shape) -> np.ndarray:
    """float32 standard normal (Box-Muller on two hashed uniform streams)."""
    n = int(np.prod(shape))
    u1 = det_u01(tag, n, stream=1)
    u2 = det_u01(tag, n, stream=2)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


def step_noise_tag(base: str, call_index: int) -> str:
    """Tag of the call_index-th normal draw of a sampling loop (0 = x_T, k = k-th randn_like)."""
    return f"{base}/draw{call_index:05d}"
