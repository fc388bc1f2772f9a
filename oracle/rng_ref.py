"""numpy replica of the device counter-hash in egt_amd/csrc/egt_common.h.

TEST INFRASTRUCTURE ONLY (see oracle/egt_oracle.py header).  The reference
samples its random attention mask with tf.random.uniform (egt_layers.py:103-106)
whose Philox stream cannot be reproduced here; the HIP kernels draw from a
counter hash on (seed, b, l, m, h) instead.  Integer arithmetic -> the device
stream must match this file bit-exactly (tests/test_rng.py).
"""
import numpy as np

DROPOUT_STREAM = np.uint32(0x5BD1E995)


def _fmix(x):
    x = x ^ (x >> np.uint32(16))
    x = x * np.uint32(0x7FEB352D)
    x = x ^ (x >> np.uint32(15))
    x = x * np.uint32(0x846CA68B)
    x = x ^ (x >> np.uint32(16))
    return x


def hash32(idx, seed, stream=0):
    """idx: uint32 array; seed: python int (64-bit)."""
    s0 = np.uint32(seed & 0xFFFFFFFF)
    s1 = np.uint32((seed >> 32) & 0xFFFFFFFF) ^ np.uint32(stream)
    with np.errstate(over="ignore"):
        x = _fmix(idx.astype(np.uint32) ^ s0)
        x = _fmix(x + s1)
    return x


def threshold24(p):
    return np.uint32(int(np.floor(float(p) * 16777216.0)))


def random_mask(seed, B, N, H, p):
    """True where the key is randomly masked (uniform < p)."""
    idx = np.arange(B * N * N * H, dtype=np.uint64).astype(np.uint32)
    u = hash32(idx, seed) >> np.uint32(8)
    return (u < threshold24(p)).reshape(B, N, N, H)


def dropout_keep(seed, B, N, H, rate):
    idx = np.arange(B * N * N * H, dtype=np.uint64).astype(np.uint32)
    u = hash32(idx, seed, DROPOUT_STREAM) >> np.uint32(8)
    return (u >= threshold24(rate)).reshape(B, N, N, H)
