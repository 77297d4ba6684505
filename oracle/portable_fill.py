"""Portable deterministic tensor fill.  TEST INFRASTRUCTURE ONLY.

A counter-based generator (splitmix64 of (crc32(name), index)) in plain numpy
integer arithmetic, so the same named tensor can be re-created bit-identically
in the build container (where the reference is imported to make golden vectors,
``oracle/make_golden.py``) and on the GPU box (where it is not present).
Golden fixtures therefore hold only OUTPUTS; inputs and weights are
regenerated from names.  Do not change the arithmetic: every committed fixture
under ``tests/golden/`` depends on it.

``fill_state_dict`` produces a numerically *conditioned* model state
(SURVEY.md F7): InstanceNorm affine parameters are perturbed away from (1, 0)
and the attention gammas are non-zero, so the generator is not the
rounding-noise amplifier it is at default init.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(name, shape, lo=-1.0, hi=1.0):
    """float64 array of ``shape`` with i.i.d.-looking U[lo, hi) values keyed by ``name``."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode("utf-8")))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        bits = _splitmix64(_splitmix64(seed) ^ (idx * np.uint64(0xD1342543DE82EF95) & _M64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def normal(name, shape):
    """Approximately N(0,1) (sum of 4 uniforms, variance-matched); exactness is irrelevant,
    determinism is the point."""
    s = sum(uniform("%s#%d" % (name, i), shape, -1.0, 1.0) for i in range(4))
    return s * np.sqrt(3.0 / 4.0)


_IN_MARKERS = (".in1.", ".in2.", "in_layers.")


def fill_value(key, shape, tag=""):
    """Conditioned value for the state-dict entry ``key`` (reference key names)."""
    name = tag + key
    if key.endswith("weight_u") or key.endswith("weight_v"):
        v = uniform(name, shape)
        return v / np.linalg.norm(v)
    if key.endswith("gamma"):
        return 0.1 * (1.0 + np.abs(uniform(name, shape)))
    is_in = any(m in ("." + key) for m in _IN_MARKERS)
    if is_in and key.endswith("weight"):
        return 1.0 + 0.3 * uniform(name, shape)
    if is_in and key.endswith("bias"):
        return 0.3 * uniform(name, shape)
    if key.endswith("bias"):
        return 0.1 * uniform(name, shape)
    if key.endswith("weight_orig") or key.endswith("weight"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        a = np.sqrt(3.0 / fan_in) * 1.4
        return a * uniform(name, shape)
    raise KeyError("no fill rule for %s" % key)


def fill_state_dict(keys_and_shapes, tag=""):
    """{key: float64 ndarray} for an ordered iterable of (key, shape)."""
    return {k: fill_value(k, tuple(s), tag) for k, s in keys_and_shapes}
