"""Checker for ``xops.spatial_dropout`` (reference models/util/extra_ops.py:137-151) -- TEST INFRASTRUCTURE, never
imported by the product (``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg only).

PARITY UNPINNED: TensorFlow's random stream cannot be reproduced (TF is not installable here and the reference holds no
fixture), so what is restated is the arithmetic of ``tf.nn.dropout(x, rate, noise_shape=[N,1,1,C])`` in TF 1.13 --
``keep_prob = 1 - rate; binary = floor(keep_prob + uniform[0,1)); y = (x / keep_prob) * binary`` with one draw per
(image, channel) plane -- and the repository's own seeded draw (include/ssal_enet.h: ssal_spatial_dropout): splitmix64
of ``seed ^ ((n*C + c) * 0xD1342543DE82EF95)``, top 24 bits / 2^24.  Written with plain Python integers so that it
shares no code with the product."""
import numpy as np

_M = (1 << 64) - 1


def _splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def keep_mask(n, c, drop_rate, seed=0):
    """float32 [n, c] of 0/1: floor((1 - rate) + u[n, c])"""
    keep_prob = np.float32(1.0) - np.float32(drop_rate)
    out = np.empty((n, c), dtype=np.float32)
    for i in range(n * c):
        h = _splitmix64((seed ^ ((i * 0xD1342543DE82EF95) & _M)) & _M)
        u = np.float32(h >> 40) * np.float32(1.0 / 16777216.0)
        out[i // c, i % c] = np.floor(np.float32(keep_prob + u))
    return out


def spatial_dropout(x, drop_rate, seed=0):
    """(x / keep_prob) * keep[n, 1, 1, c] in float32"""
    x = np.asarray(x, dtype=np.float32)
    n, _, _, c = x.shape
    keep_prob = np.float32(1.0) - np.float32(drop_rate)
    return (x / keep_prob) * keep_mask(n, c, drop_rate, seed)[:, None, None, :]
