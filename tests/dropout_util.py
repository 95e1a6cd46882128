"""numpy replica of the engine's counter-based dropout mask (csrc/common.h dropout_bits; pinned to the
library's own host evaluation, pinn_dropout_keep, in tests/test_host_cpu.py)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _fmix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & M32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & M32
    x ^= x >> np.uint64(16)
    return x


def dropout_bits(seed, layer, feature, point):
    """32 uniform bits per (seed, hidden layer, unit, point); arrays broadcast."""
    point = np.asarray(point, np.uint64)
    feature = np.asarray(feature, np.uint64)
    x = _fmix(np.uint64(seed) ^ ((point * np.uint64(0x9E3779B1)) & M32))
    y = (((np.uint64(layer) * np.uint64(0x01000193) + feature) & M32) * np.uint64(0x9E3779B1) + (point >> np.uint64(32))) & M32
    return _fmix(x ^ y)


def keep_masks(seed, p, n_hidden, width, N):
    """One (N, width) 0/1 float32 array per hidden layer: the mask the kernels apply under (seed, p)."""
    thresh = min(int(float(np.float32(p)) * 4294967296.0), 4294967295)
    thresh = max(thresh, 1) if p > 0 else 0
    pts = np.arange(N, dtype=np.uint64)[:, None]
    feats = np.arange(width, dtype=np.uint64)[None, :]
    return [(dropout_bits(seed, l, feats, pts) >= np.uint64(thresh)).astype(np.float32) for l in range(n_hidden)]
