"""Closed-form test tensors: large fixtures without large files.

The full-width fixtures (F10 at ch=8, F11 at ch=32) would need tens to hundreds of MB of initial weights and input
clips.  Instead every LARGE tensor of those fixtures is defined by an integer hash of (name, element index) -- exact
integer arithmetic in numpy, so make_golden.py (which installs the values into the real reference before running
it) and the tests (which install the same values into the oracle / the HIP modules) construct bit-identical arrays
on any machine.  Small tensors (biases, SN u / v, BN buffers, embeddings) are stored verbatim in the .npz.
"""
import zlib

import numpy as np

BIG = 4096          # tensors with at least this many elements are synthesised, smaller ones are stored


def uniform(name, shape, scale=1.0):
    """float32 array of `shape`, i.i.d.-looking uniform in [-scale, scale): splitmix64 of (crc32(name), index)."""
    n = int(np.prod(shape))
    x = np.arange(n, dtype=np.uint64) + (np.uint64(zlib.crc32(name.encode())) << np.uint64(32))
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    u = (x >> np.uint64(40)).astype(np.float64) / float(1 << 24)            # 24 bits -> [0, 1), exact in float32
    return ((u * 2.0 - 1.0) * scale).astype(np.float32).reshape(shape)


def weight(name, shape):
    """Synthetic initial weight: uniform with the variance of torch's default conv / linear init (bound 1/sqrt(fan_in))."""
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    return uniform(name, shape, 1.0 / np.sqrt(fan_in))


def fill_state(stored, template, tag):
    """Complete state_dict for network `tag`: entries present in `stored` (keys '<tag>.sd0.<key>') verbatim, every other
    floating-point entry of `template` ({key: shape}) synthesised.  Returns {key: np.ndarray}."""
    out = {}
    pfx = tag + ".sd0."
    for key, shape in template.items():
        if pfx + key in stored:
            out[key] = stored[pfx + key]
        else:
            out[key] = weight(tag + "." + key, tuple(shape))
    return out
