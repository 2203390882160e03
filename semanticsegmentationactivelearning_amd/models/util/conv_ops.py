"""Stand-alone forms of the operators the ICNet row is built from (``include/ssal_icnet.h``), on GPU-resident
torch tensors (NHWC fp32).  The reference has no counterpart module (its ``models/icnet/icnet.py:1-7`` is empty);
semantics are those ``ICNET_SPEC.md`` cites: SAME convolutions as ``models/enet/enet_modules.py:205,538,565,581``,
batch-norm ``models/util/extra_ops.py:154-185``, bilinear resize ``inference.py:96-99``.  No CPU fallback.
"""
import ctypes

import numpy as np

from ... import _lib


def _host(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _hp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def conv_bn_act(x, kernel, stride=1, dilation=1, bn=None, bias=None, residual=None, relu=True, upsample2x=False):
    """``[relu]( BN(conv2d(x, kernel HWIO, strides, dilations, "SAME")) [+ residual] )`` in ONE launch on the fp32
    matrix cores (``cin % 32 == 0``), or the 3x3 / stride-2 first-layer kernel (``cin`` in 1, 3, 4 -> 32 channels).
    ``bn`` = (mean, variance, gamma, beta) numpy vectors or None; ``upsample2x`` runs the convolution on
    ``tf.image.resize_bilinear(x, 2x)`` evaluated on the fly (cascade feature fusion, ICNET_SPEC section 4)."""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    k = _host(kernel)
    kh, kw, cin, cout = k.shape
    n, h, w, c = x.shape
    if c != cin:
        raise ValueError("kernel expects %d input channels, got %d" % (cin, c))
    m = v = g = b = None
    if bn is not None:
        m, v, g, b = (_host(t) for t in bn)
    bias = _host(bias)
    hh, ww = (2 * h, 2 * w) if upsample2x else (h, w)
    oh, ow = -(-hh // stride), -(-ww // stride)
    res = None
    if residual is not None:
        res = _lib.as_device_f32(residual)
        if tuple(res.shape) != (n, oh, ow, cout):
            raise ValueError("residual must have shape %s (got %s)" % ((n, oh, ow, cout), tuple(res.shape)))
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.ssal_conv_bn_workspace_bytes(kh, kw, cin, cout)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
        y = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=x.device)
        _lib.check(L.ssal_conv_bn_act(_lib.dev_ptr(x), n, h, w, cin, _hp(k), kh, kw, cout, int(stride), int(dilation),
                                      _hp(m), _hp(v), _hp(g), _hp(b), _hp(bias), _lib.dev_ptr(res), 1 if relu else 0,
                                      1 if upsample2x else 0, _lib.dev_ptr(y), _lib.dev_ptr(ws), ws.numel(),
                                      _lib.stream_ptr()))
    return y


def max_pool_3x3_s2(x):
    """``tf.nn.max_pool(x, 3x3, strides 2, "SAME")``"""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssal_max_pool_3x3_s2(_lib.dev_ptr(x), n, h, w, c, _lib.dev_ptr(y), _lib.stream_ptr()))
    return y


def pyramid_pooling(x):
    """ICNET_SPEC conv5_3_sum: ``x + sum_b resize_bilinear(bin_average_b(x))`` for b in (1, 2, 3, 6)"""
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(x)
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        ws = torch.empty(n * (50 + 12 * h) * c * 4, dtype=torch.uint8, device=x.device)
        _lib.check(_lib.lib().ssal_pyramid_pooling(_lib.dev_ptr(x), n, h, w, c, _lib.dev_ptr(y), _lib.dev_ptr(ws),
                                                   ws.numel(), _lib.stream_ptr()))
    return y


def upscore_logits(logits_quarter, measure="margin", threshold=0.0, return_label=False, return_mask=False,
                   return_confidence=False):
    """conv6_interp (4x ``resize_bilinear``) + softmax + acquisition measure + float64 per-image mean on
    materialised 1/4-resolution logits ``[N,h,w,classes]`` -> scores [N] (+ optional [N,4h,4w] maps)."""
    if measure not in _lib.MEASURES:
        raise NotImplementedError("Uncertainty function not implemented.")
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(logits_quarter)
    n, h, w, k = x.shape
    L = _lib.lib()
    with torch.cuda.device(x.device):
        ws = torch.empty(int(L.ssal_upscore_workspace_bytes(n, h, w)), dtype=torch.uint8, device=x.device)
        scores = torch.empty((n,), dtype=torch.float64, device=x.device)
        label = torch.empty((n, 4 * h, 4 * w), dtype=torch.uint8, device=x.device) if return_label else None
        mask = torch.empty((n, 4 * h, 4 * w), dtype=torch.uint8, device=x.device) if return_mask else None
        conf = torch.empty((n, 4 * h, 4 * w), dtype=torch.float32, device=x.device) if return_confidence else None
        _lib.check(L.ssal_upscore_logits_nhwc(_lib.dev_ptr(x), n, h, w, k, _lib.MEASURES[measure], float(threshold),
                                              _lib.dev_ptr(scores), _lib.dev_ptr(label), _lib.dev_ptr(mask),
                                              _lib.dev_ptr(conf), _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr()))
    if return_label or return_mask or return_confidence:
        return scores, {"label": label, "mask": mask, "confidence": conf}
    return scores
