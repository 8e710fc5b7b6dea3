"""Deterministic, library-independent test-vector generator.

Used by BOTH `make_golden.py` (which runs the reference in the dev container)
and the tests (which rebuild the very same inputs on the GPU box, where
`/root/reference` does not exist).  Everything is integer arithmetic on
numpy uint64 (splitmix64), so the stream does not depend on numpy's or
torch's RNG implementation or version.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def det_u01(shape, seed):
    """float64 uniforms in [0,1) from a counter-based hash; exact everywhere."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        bits = _splitmix64(_splitmix64(ctr))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def det_uniform(shape, seed, lo=-1.0, hi=1.0):
    return (lo + (hi - lo) * det_u01(shape, seed)).astype(np.float32)


def det_normal(shape, seed):
    """Standard normals via Box-Muller on two hashed streams (float32 result)."""
    u1 = det_u01(shape, seed * 2 + 1)
    u2 = det_u01(shape, seed * 2 + 2)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def det_bernoulli(shape, seed, p):
    return det_u01(shape, seed) < p


def linear_init(out_f, in_f, seed, gain=1.0):
    """U(-g/sqrt(in), g/sqrt(in)) weight + bias for a Linear(in_f, out_f)."""
    b = gain / np.sqrt(in_f)
    return det_uniform((out_f, in_f), seed, -b, b), det_uniform((out_f,), seed + 7919, -b, b)
