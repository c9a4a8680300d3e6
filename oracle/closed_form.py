"""Closed-form deterministic parameter values (test infrastructure).

Fresh init is useless for parity: ``zero_module`` (reference nn.py:516-522)
zeroes the last conv of every ResBlock, every attention ``proj_out`` and the
output head, so a freshly built reference model outputs exactly 0.  Instead
every tensor of a state dict is filled from a hash of (key name, flat index):
platform independent (numpy uint64 wrap-around arithmetic, no torch RNG) and
independent of parameter registration order.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def uniform_pm1(n, seed):
    """n float64 values in [-1, 1), a pure function of (seed, index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        s = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(0x51ED))
        bits = _splitmix64(idx ^ s)
    return (bits >> np.uint64(11)).astype(np.float64) * (2.0 / (1 << 53)) - 1.0


def key_seed(key, salt=0):
    return (zlib.crc32(key.encode()) + 0x9E37 * salt) & 0xFFFFFFFF


def fill_value(key, shape, salt=0):
    """Value for state-dict entry `key` of `shape` (logical, row-major order)."""
    shape = tuple(int(s) for s in shape)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.int64)
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform_pm1(n, key_seed(key, salt))
    if key.endswith("running_var"):
        v = 1.0 + 0.25 * u
    elif key.endswith("running_mean"):
        v = 0.1 * u
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(3.0 / fan_in)
    elif key.endswith("weight"):
        v = 1.0 + 0.2 * u          # norm gains
    else:
        v = 0.1 * u                # biases
    return torch.from_numpy(v.reshape(shape).astype(np.float32))


def fill_state_dict(spec, salt=0):
    """spec: iterable of (key, shape) -> OrderedDict key -> tensor."""
    return OrderedDict((k, fill_value(k, s, salt)) for k, s in spec)


def synth(name, shape, lo=-1.0, hi=1.0):
    """Deterministic synthetic input tensor (same hash family, by name)."""
    n = int(np.prod(shape))
    u = uniform_pm1(n, key_seed("input:" + name))
    v = lo + (u + 1.0) * 0.5 * (hi - lo)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))
