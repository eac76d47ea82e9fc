"""Philox4x32-10 counter-based RNG (Salmon et al., SC'11) in numpy -- test oracle.

The reference samples with unseeded TF RNG (utils/training.py:110,115), which cannot be
bit-matched.  The HIP sampler and this oracle share this counter-based generator so that the
sampled indices are reproducible and comparable; tests additionally check the distributional
contract of the reference.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(counter, key, rounds=10):
    """counter: uint32[..., 4], key: (k0, k1) python ints.  Returns uint32[..., 4]."""
    c = np.asarray(counter, dtype=np.uint64) & _MASK
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(rounds):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def rand_u32(i, image, step, stream, seed):
    """First output word for counter (i, image, step, stream) under key = 64-bit seed."""
    i = np.atleast_1d(np.asarray(i, dtype=np.uint64))
    ctr = np.zeros(i.shape + (4,), dtype=np.uint64)
    ctr[..., 0] = i
    ctr[..., 1] = image
    ctr[..., 2] = step
    ctr[..., 3] = stream
    out = philox4x32(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return out[..., 0]
