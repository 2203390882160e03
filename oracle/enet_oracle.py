"""TEST INFRASTRUCTURE ONLY -- parity oracle, never imported by the product path.

ENet forward + acquisition scoring restated on the CPU: the network wiring follows the reference
(models/enet/enet.py:35-247, 320-367; models/enet/enet_modules.py call methods :190-224, :526-599,
:868-938, :1217-1292, :1359-1381; active_learning.py:229-263, 682-715) and every tensor op is
executed by the plain-C restatement in ``enet_oracle.c`` (fixed fmaf accumulation order, see its
header).  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module.

PARITY STATUS: "parity unpinned" against real TensorFlow 1.13 -- TensorFlow is not installable here
and the reference ships no golden vectors for this path.  Pinned by: the reference's
pool->unpool->pool invariant (models/util/test_xops.py:6-21), agreement with the independent
torch-CPU restatement (oracle/torch_restatement.py) and the committed fixtures in tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libenet_oracle.so")
_LIB = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """gcc-compile the C restatement (content-stamped: mtimes do not survive a snapshot copy)"""
    import hashlib
    digest = hashlib.sha256(b"".join(open(os.path.join(_HERE, f), "rb").read()
                                     for f in ("enet_oracle.c", "icnet_oracle.c", "Makefile"))).hexdigest()
    stamp = _SO + ".stamp"
    fresh = os.path.exists(_SO) and os.path.exists(stamp) and open(stamp).read().strip() == digest
    if force or not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libenet_oracle.so"], stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(digest)
    return _SO


def usable_cores(cap=None):
    """CPU threads this process may really use: min(affinity, cgroup quota[, cap]) -- the oracle's own copy (test
    infrastructure imports nothing from the product package)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    env = os.environ.get("SSAL_CPU_THREADS")
    if env:
        n = min(n, max(1, int(env)))
    if cap:
        n = min(n, cap)
    return max(1, n)


def _lib():
    global _LIB
    if _LIB is None:
        build()
        if "OMP_NUM_THREADS" not in os.environ:  # size the OpenMP pool to the box's real CPU share
            os.environ["OMP_NUM_THREADS"] = str(usable_cores(cap=32))
        _LIB = ctypes.CDLL(_SO)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- primitive ops (each a thin wrapper over the C restatement) ------------------------------------
def bn_fold(mean, var, gamma, beta):
    c = len(mean)
    s, t = np.empty(c, np.float32), np.empty(c, np.float32)
    _lib().orc_bn_fold(_p(_f32(mean)), _p(_f32(var)), _p(_f32(gamma)), _p(_f32(beta)), c, _p(s), _p(t))
    return s, t


def conv2d_same(x, w, stride=1, dil=1):
    x, w = _f32(x), _f32(w)
    n, h, ww, cin = x.shape
    kh, kw, ci, co = w.shape
    assert ci == cin, (w.shape, x.shape)
    ho, wo = -(-h // stride), -(-ww // stride)
    y = np.empty((n, ho, wo, co), np.float32)
    _lib().orc_conv2d_same(_p(x), n, h, ww, cin, _p(w), kh, kw, co, stride, dil, _p(y))
    return y


def conv2d_transpose_3x3_s2(x, w):
    x, w = _f32(x), _f32(w)
    n, h, ww, cin = x.shape
    assert w.shape[:2] == (3, 3) and w.shape[3] == cin, (w.shape, x.shape)
    co = w.shape[2]
    y = np.empty((n, 2 * h, 2 * ww, co), np.float32)
    _lib().orc_conv2d_transpose_3x3_s2(_p(x), n, h, ww, cin, _p(w), co, _p(y))
    return y


def affine_prelu(x, scale=None, shift=None, alpha=None):
    x = _f32(x)
    c = x.shape[-1]
    y = np.empty_like(x)
    lib = _lib()
    lib.orc_affine_prelu.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    s = _f32(scale) if scale is not None else None
    t = _f32(shift) if shift is not None else None
    a = _f32(alpha) if alpha is not None else None
    lib.orc_affine_prelu(_p(x), x.size // c, c, _p(s) if s is not None else None,
                         _p(t) if t is not None else None, _p(a) if a is not None else None, _p(y))
    return y


def add_prelu(a, res, alpha):
    a, res, alpha = _f32(a), _f32(res), _f32(alpha)
    c, cres = a.shape[-1], res.shape[-1]
    y = np.empty_like(a)
    lib = _lib()
    lib.orc_add_prelu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.orc_add_prelu(_p(a), _p(res), a.size // c, c, cres, _p(alpha), _p(y))
    return y


def maxpool2x2_argmax(x, include_batch=False):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, h // 2, w // 2, c), np.float32)
    idx = np.empty((n, h // 2, w // 2, c), np.int64)
    _lib().orc_maxpool2x2_argmax(_p(x), n, h, w, c, _p(y), _p(idx), 1 if include_batch else 0)
    return y, idx


def unpool2d(x, idx, idx_has_batch=False):
    x = _f32(x)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n, h, w, c = x.shape
    y = np.empty((n, 2 * h, 2 * w, c), np.float32)
    _lib().orc_unpool2d(_p(x), _p(idx), n, h, w, c, 1 if idx_has_batch else 0, _p(y))
    return y


def concat2(a, b):
    a, b = _f32(a), _f32(b)
    y = np.empty(a.shape[:-1] + (a.shape[-1] + b.shape[-1],), np.float32)
    lib = _lib()
    lib.orc_concat2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_size_t, ctypes.c_void_p]
    lib.orc_concat2(_p(a), a.shape[-1], _p(b), b.shape[-1], a.size // a.shape[-1], _p(y))
    return y


MEASURES = {"entropy": 0, "margin": 1, "confidence": 2}


def score_logits(logits, measure="entropy"):
    """-> (mean float64 [N], per-pixel confidence float32 [N,H,W], label uint8 [N,H,W])
    reference active_learning.py:234-263."""
    if measure not in MEASURES:
        raise NotImplementedError("Uncertainty function not implemented.")
    x = _f32(logits)
    n, h, w, k = x.shape
    conf = np.empty((n, h, w), np.float32)
    label = np.empty((n, h, w), np.uint8)
    mean = np.empty((n,), np.float64)
    rc = _lib().orc_score(_p(x), n, h, w, k, MEASURES[measure], _p(conf), _p(label), _p(mean))
    assert rc == 0
    return mean, conf, label


# ---- blocks ----------------------------------------------------------------------------------------
def _bn(P, prefix):
    return bn_fold(P[prefix + "mean"], P[prefix + "variance"], P[prefix + "gamma"], P[prefix + "beta"])


def initial(P, name, x):
    """enet_modules.py:190-224"""
    conv = conv2d_same(x, P[name + ".kernel"], stride=2)
    pool, _ = maxpool2x2_argmax(x)
    s, t = _bn(P, name + ".")
    return affine_prelu(concat2(conv, pool), s, t, P[name + ".alpha"])


def bottleneck(P, name, x, dil=1, asym=False):
    """enet_modules.py:526-599"""
    s, t = _bn(P, name + ".proj_")
    y = affine_prelu(conv2d_same(x, P[name + ".proj_kernel"]), s, t, P[name + ".proj_alpha"])
    if asym:
        y = conv2d_same(y, P[name + ".conv_kernel.0"])
        y = conv2d_same(y, P[name + ".conv_kernel.1"])
    else:
        y = conv2d_same(y, P[name + ".conv_kernel"], dil=dil)
    s, t = _bn(P, name + ".conv_")
    y = affine_prelu(y, s, t, P[name + ".conv_alpha"])
    s, t = _bn(P, name + ".exp_")
    y = affine_prelu(conv2d_same(y, P[name + ".exp_kernel"]), s, t, None)
    return add_prelu(y, x, P[name + ".residual_alpha"])


def bottleneck_down(P, name, x):
    """enet_modules.py:868-938 -> (output, argmax int64 per-image index)"""
    s, t = _bn(P, name + ".proj_")
    y = affine_prelu(conv2d_same(x, P[name + ".proj_kernel"], stride=2), s, t, P[name + ".proj_alpha"])
    s, t = _bn(P, name + ".conv_")
    y = affine_prelu(conv2d_same(y, P[name + ".conv_kernel"]), s, t, P[name + ".conv_alpha"])
    s, t = _bn(P, name + ".exp_")
    y = affine_prelu(conv2d_same(y, P[name + ".exp_kernel"]), s, t, None)
    pool, argmax = maxpool2x2_argmax(x)
    return add_prelu(y, pool, P[name + ".residual_alpha"]), argmax


def bottleneck_up(P, name, x, argmax):
    """enet_modules.py:1217-1292"""
    s, t = _bn(P, name + ".proj_")
    y = affine_prelu(conv2d_same(x, P[name + ".proj_kernel"]), s, t, P[name + ".proj_alpha"])
    s, t = _bn(P, name + ".conv_")
    y = affine_prelu(conv2d_transpose_3x3_s2(y, P[name + ".conv_kernel"]), s, t, P[name + ".conv_alpha"])
    s, t = _bn(P, name + ".exp_")
    y = affine_prelu(conv2d_same(y, P[name + ".exp_kernel"]), s, t, None)
    res = unpool2d(conv2d_same(x, P[name + ".res_kernel"]), argmax)
    return add_prelu(y, res, P[name + ".residual_alpha"])


def final(P, name, x):
    """enet_modules.py:1359-1381"""
    return conv2d_transpose_3x3_s2(x, P[name + ".kernel"])


_STAGE23 = [(1, 1, False), (2, 2, False), (3, 1, True), (4, 4, False), (5, 1, False), (6, 8, False),
            (7, 1, True), (8, 16, False)]


def enet_forward(P, x, endpoints=None):
    """ENet.call(inputs, training=False) (models/enet/enet.py:320-367).  ``endpoints`` (dict) collects
    every block output by layer name plus "argmax1"/"argmax2"."""
    ep = endpoints if endpoints is not None else {}
    y = ep["Initial"] = initial(P, "Initial", x)
    y, a1 = bottleneck_down(P, "Bottleneck1_0", y)
    ep["Bottleneck1_0"], ep["argmax1"] = y, a1
    for i in range(1, 5):
        y = ep["Bottleneck1_%d" % i] = bottleneck(P, "Bottleneck1_%d" % i, y)
    y, a2 = bottleneck_down(P, "Bottleneck2_0", y)
    ep["Bottleneck2_0"], ep["argmax2"] = y, a2
    for stage in (2, 3):
        for i, dil, asym in _STAGE23:
            nm = "Bottleneck%d_%d" % (stage, i)
            y = ep[nm] = bottleneck(P, nm, y, dil=dil, asym=asym)
    y = ep["Bottleneck4_0"] = bottleneck_up(P, "Bottleneck4_0", y, a2)
    y = ep["Bottleneck4_1"] = bottleneck(P, "Bottleneck4_1", y)
    y = ep["Bottleneck4_2"] = bottleneck(P, "Bottleneck4_2", y)
    y = ep["Bottleneck5_0"] = bottleneck_up(P, "Bottleneck5_0", y, a1)
    y = ep["Bottleneck5_1"] = bottleneck(P, "Bottleneck5_1", y)
    y = ep["Final"] = final(P, "Final", y)
    return y


def score_images(P, x, measure="entropy"):
    """forward + score: (mean float64 [N], conf [N,H,W], label [N,H,W], logits)"""
    logits = enet_forward(P, x)
    mean, conf, label = score_logits(logits, measure)
    return mean, conf, label, logits


def rank_lowest(scores_f64_by_example, unlabelled, selection_size):
    """Host tail of rank_confidence (active_learning.py:685,700,705-715)."""
    confidence = np.asarray(scores_f64_by_example).astype(np.float32)
    unl = np.asarray(unlabelled, dtype=np.int64)
    uc = confidence[unl]
    k = int(np.minimum(len(unl), selection_size))
    if k >= len(uc):
        return unl.copy(), uc
    return unl[np.argpartition(uc, k)[:k]], uc


def masked_softmax_cross_entropy(labels, logits, mask, num_classes, weight=0.0, label_smoothing=0.0):
    """tensortools/losses.py:3-74 (forward value, float64)"""
    lg = _f32(logits)
    n, h, w, k = lg.shape
    assert k == num_classes
    lab = np.ascontiguousarray(np.asarray(labels).reshape(n, h, w), dtype=np.uint8)
    mk = _f32(np.asarray(mask).reshape(n, h, w))
    lib = _lib()
    lib.orc_masked_softmax_xent.restype = ctypes.c_double
    lib.orc_masked_softmax_xent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    return float(lib.orc_masked_softmax_xent(_p(lg), _p(lab), _p(mk), n, h, w, k, weight, label_smoothing))


def resize_nearest_neighbor(x, size):
    """tf.image.resize_nearest_neighbor, align_corners=False: src = floor(dst * in/out) (float32), clamped"""
    x = np.asarray(x)
    h, w = x.shape[1], x.shape[2]
    oh, ow = int(size[0]), int(size[1])
    iy = np.minimum(np.floor(np.arange(oh, dtype=np.float32) * (np.float32(h) / np.float32(oh))).astype(np.int64), h - 1)
    ix = np.minimum(np.floor(np.arange(ow, dtype=np.float32) * (np.float32(w) / np.float32(ow))).astype(np.int64), w - 1)
    return x[:, iy][:, :, ix]


def multiscale_masked_softmax_cross_entropy(labels, logits, mask, num_classes, kernels, weight=0.0,
                                            label_smoothing=0.0):
    """tensortools/losses.py:76-157 (forward value): logits[0] = class logits, logits[1:] = lower-scale
    features, each with its 1x1 head kernels[i] [1,1,C,K]; labels / mask nearest-neighbour resized"""
    n, h, w, _ = np.asarray(logits[0]).shape
    lab = np.asarray(labels).reshape(n, h, w)
    mk = np.asarray(mask, dtype=np.float32).reshape(n, h, w)
    total = masked_softmax_cross_entropy(lab, logits[0], mk, num_classes, weight, label_smoothing)
    for feat, krnl in zip(logits[1:], kernels):
        head = conv2d_same(_f32(feat), _f32(krnl), 1, 1)
        size = head.shape[1:3]
        total += masked_softmax_cross_entropy(resize_nearest_neighbor(lab, size), head,
                                              resize_nearest_neighbor(mk, size), num_classes, weight,
                                              label_smoothing)
    return total
