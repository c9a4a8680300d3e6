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


ZERO_INIT_SUFFIXES = ("out_layers.3.weight", "out_layers.3.bias", "proj_out.weight", "proj_out.bias")


def fill_value_trained(key, shape, salt=0):
    """"Trained-like" value for state-dict entry `key`: matrices / conv kernels get LOG-UNIFORM magnitudes over four decades with
    a random sign, |w| = 10^(-4 v) * g / sqrt(fan_in), v uniform in [0, 1) — heavy-tailed and mostly small, the shape of a trained
    network's weight histogram (the uniform fill of fill_value never leaves one binade and a half).  g = 4.292 makes the variance
    1 / fan_in like fill_value's.  A quarter of the reference's zero-initialised layers (nn.py:516-522 zero_module: the last conv of a
    ResBlock, the attention proj_out) are LEFT ZERO, chosen by a hash of the key: a partially trained network, and the all-zero
    tensor case of the weight-scale records.  Vectors (norm gains, biases) as in fill_value."""
    shape = tuple(int(s) for s in shape)
    if len(shape) < 2:
        return fill_value(key, shape, salt)
    if key.endswith(ZERO_INIT_SUFFIXES[0]) or key.endswith(ZERO_INIT_SUFFIXES[2]):
        if zlib.crc32(key.encode()) % 4 == 0:
            return torch.zeros(shape, dtype=torch.float32)
    n = int(np.prod(shape))
    u = uniform_pm1(n, key_seed(key, salt))                       # sign
    v = (uniform_pm1(n, key_seed(key, salt + 7)) + 1.0) * 0.5     # exponent
    fan_in = int(np.prod(shape[1:]))
    g = 1.0 / np.sqrt((1.0 - 1e-8) / (8.0 * np.log(10.0)))       # 1 / rms of 10^(-4 v)
    w = np.where(u < 0, -1.0, 1.0) * np.power(10.0, -4.0 * v) * (g / np.sqrt(fan_in))
    return torch.from_numpy(w.reshape(shape).astype(np.float32))


def fill_state_dict_trained(spec, salt=0):
    return OrderedDict((k, fill_value_trained(k, s, salt)) for k, s in spec)


def synth_noise(name, shape):
    """Closed-form stand-in for a unit-variance noise draw (uniform in +-sqrt(3)): both sides of a fixture generate it, so a 1000-step
    ancestral loop needs no stored noise."""
    r = float(np.sqrt(3.0))
    return synth(name, shape, -r, r)


def fill_state_dict(spec, salt=0):
    """spec: iterable of (key, shape) -> OrderedDict key -> tensor."""
    return OrderedDict((k, fill_value(k, s, salt)) for k, s in spec)


def synth(name, shape, lo=-1.0, hi=1.0):
    """Deterministic synthetic input tensor (same hash family, by name)."""
    n = int(np.prod(shape))
    u = uniform_pm1(n, key_seed("input:" + name))
    v = lo + (u + 1.0) * 0.5 * (hi - lo)
    return torch.from_numpy(v.reshape(shape).astype(np.float32))
