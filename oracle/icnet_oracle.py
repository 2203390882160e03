"""TEST INFRASTRUCTURE ONLY -- parity oracle of the ICNet row, never imported by the product path.

CPU restatement of ``ICNET_SPEC.md`` (the reference's ``models/icnet/icnet.py:1-7`` is an empty class whose
docstring cites the ICNet paper; the spec pins the architecture and expresses every operator with the semantics the
reference repository defines for it: SAME convolutions as in ``models/enet/enet_modules.py``, batch-norm
``models/util/extra_ops.py:154-185``, bilinear resize ``inference.py:96-99``, acquisition measures
``active_learning.py:239-263``).  Every tensor op runs in the plain-C restatements ``enet_oracle.c`` /
``icnet_oracle.c`` (fixed fmaf accumulation order).

PARITY STATUS: "parity unpinned AND undefined" -- there is no reference behaviour for this network.  Pinned only by
agreement with the independent torch-CPU restatement (``torch_restatement.icnet_forward``) and the committed
self-generated fixture ``tests/golden/icnet_c3k19_64x128.npz``.
"""
import ctypes

import numpy as np

from . import enet_oracle as eo
from .enet_oracle import _f32, _lib, _p

PPM_BINS = (1, 2, 3, 6)


def conv_specs(c_in=3, classes=19):
    """[(name, k, cin, cout, stride, dilation)] of every convolution, ICNET_SPEC sections 1-4"""
    L = []

    def conv(name, k, cin, cout, s=1, d=1):
        L.append((name, k, cin, cout, s, d))

    def bneck(name, cin, mid, cout, s, d, proj):
        conv(name + "_1x1_reduce", 1, cin, mid, s)
        conv(name + "_3x3", 3, mid, mid, 1, d)
        conv(name + "_1x1_increase", 1, mid, cout)
        if proj:
            conv(name + "_1x1_proj", 1, cin, cout, s)

    conv("conv1_1_3x3_s2", 3, c_in, 32, 2)
    conv("conv1_2_3x3", 3, 32, 32)
    conv("conv1_3_3x3", 3, 32, 64)
    for name, cin, mid, cout, s, d, proj in BNECKS:
        bneck(name, cin, mid, cout, s, d, proj)
    conv("conv5_4_k1", 1, 1024, 256)
    conv("conv_sub4", 3, 256, 128, 1, 2)
    conv("conv3_1_sub2_proj", 1, 256, 128)
    conv("conv_sub2", 3, 128, 128, 1, 2)
    conv("conv1_sub1", 3, c_in, 32, 2)
    conv("conv2_sub1", 3, 32, 32, 2)
    conv("conv3_sub1", 3, 32, 64, 2)
    conv("conv3_sub1_proj", 1, 64, 128)
    conv("conv6_cls", 1, 128, classes)
    return L


# (name, cin, mid, cout, stride, dilation, projection shortcut) -- ICNET_SPEC sections 1 and 2
BNECKS = ([("conv2_1", 64, 32, 128, 1, 1, True), ("conv2_2", 128, 32, 128, 1, 1, False),
           ("conv2_3", 128, 32, 128, 1, 1, False), ("conv3_1", 128, 64, 256, 2, 1, True)]
          + [("conv3_%d" % i, 256, 64, 256, 1, 1, False) for i in (2, 3, 4)]
          + [("conv4_1", 256, 128, 512, 1, 2, True)]
          + [("conv4_%d" % i, 512, 128, 512, 1, 2, False) for i in (2, 3, 4, 5, 6)]
          + [("conv5_1", 512, 256, 1024, 1, 4, True)]
          + [("conv5_%d" % i, 1024, 256, 1024, 1, 4, False) for i in (2, 3)])


def param_shapes(c_in=3, classes=19):
    """{"<layer>.<attr>": shape} in the C-ABI naming"""
    out = {}
    for name, k, cin, cout, s, d in conv_specs(c_in, classes):
        out[name + ".kernel"] = (k, k, cin, cout)
        if name == "conv6_cls":
            out[name + ".bias"] = (cout,)
        else:
            for a in ("mean", "variance", "gamma", "beta"):
                out["%s.%s" % (name, a)] = (cout,)
    return out


# ---- primitive ops ---------------------------------------------------------------------------------
def affine_add_relu(x, scale=None, shift=None, res=None, relu=True):
    x = _f32(x)
    c = x.shape[-1]
    y = np.empty_like(x)
    lib = _lib()
    lib.orc_affine_add_relu.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    s = _f32(scale) if scale is not None else None
    t = _f32(shift) if shift is not None else None
    r = _f32(res) if res is not None else None
    assert r is None or r.shape == x.shape, (r.shape, x.shape)
    lib.orc_affine_add_relu(_p(x), x.size // c, c, _p(s) if s is not None else None,
                            _p(t) if t is not None else None, _p(r) if r is not None else None,
                            1 if relu else 0, _p(y))
    return y


def maxpool3x3_s2(x):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, (h + 1) // 2, (w + 1) // 2, c), np.float32)
    _lib().orc_maxpool3x3_s2_same(_p(x), n, h, w, c, _p(y))
    return y


def resize_bilinear(x, oh, ow):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, oh, ow, c), np.float32)
    _lib().orc_resize_bilinear(_p(x), n, h, w, c, int(oh), int(ow), _p(y))
    return y


def adaptive_avg_pool(x, b):
    x = _f32(x)
    n, h, w, c = x.shape
    y = np.empty((n, b, b, c), np.float32)
    _lib().orc_adaptive_avg_pool(_p(x), n, h, w, c, int(b), _p(y))
    return y


def add(a, b):
    a, b = _f32(a), _f32(b)
    assert a.shape == b.shape
    y = np.empty_like(a)
    lib = _lib()
    lib.orc_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.orc_add(_p(a), _p(b), a.size, _p(y))
    return y


# ---- blocks ----------------------------------------------------------------------------------------
def conv_bn(P, name, x, stride=1, dil=1, relu=True, res=None):
    """conv (SAME, no bias) -> folded batch-norm -> [+ res] -> [relu]"""
    y = eo.conv2d_same(x, P[name + ".kernel"], stride=stride, dil=dil)
    s, t = eo.bn_fold(P[name + ".mean"], P[name + ".variance"], P[name + ".gamma"], P[name + ".beta"])
    return affine_add_relu(y, s, t, res, relu)


def bottleneck(P, name, x, stride, dil, proj, ep=None):
    """ICNET_SPEC "Bneck": relu(BN(increase(relu(BN(3x3(relu(BN(reduce(x)))))))) + shortcut)"""
    ep = ep if ep is not None else {}
    sc = x
    if proj:
        sc = ep[name + "_1x1_proj"] = conv_bn(P, name + "_1x1_proj", x, stride=stride, relu=False)
    y = ep[name + "_1x1_reduce"] = conv_bn(P, name + "_1x1_reduce", x, stride=stride)
    y = ep[name + "_3x3"] = conv_bn(P, name + "_3x3", y, dil=dil)
    return conv_bn(P, name + "_1x1_increase", y, relu=True, res=sc)


def pyramid_pooling(x):
    """ICNET_SPEC conv5_3_pool* / conv5_3_sum: ((((x + up1) + up2) + up3) + up6)"""
    n, h, w, c = x.shape
    y = x
    for b in PPM_BINS:
        y = add(y, resize_bilinear(adaptive_avg_pool(x, b), h, w))
    return y


def cff(P, f_low, f_high, conv_name, proj_name):
    """cascade feature fusion (ICNET_SPEC section 4)"""
    n, h, w, _ = f_low.shape
    up = resize_bilinear(f_low, 2 * h, 2 * w)
    b = conv_bn(P, proj_name, f_high, relu=False)
    return conv_bn(P, conv_name, up, dil=2, relu=True, res=b), up, b


def icnet_forward(P, x, endpoints=None, full_logits=True):
    """-> logits [N,H,W,classes] (or the 1/4-resolution logits with ``full_logits=False``).
    ``endpoints`` (dict) receives every named intermediate tensor."""
    x = _f32(x)
    n, h, w, _ = x.shape
    assert h % 32 == 0 and w % 32 == 0, "ICNet needs H and W divisible by 32"
    ep = endpoints if endpoints is not None else {}

    def keep(name, t):
        ep[name] = t
        return t

    # medium-resolution branch / shared stem
    y = keep("data_sub2", resize_bilinear(x, h // 2, w // 2))
    y = keep("conv1_1_3x3_s2", conv_bn(P, "conv1_1_3x3_s2", y, stride=2))
    y = keep("conv1_2_3x3", conv_bn(P, "conv1_2_3x3", y))
    y = keep("conv1_3_3x3", conv_bn(P, "conv1_3_3x3", y))
    y = keep("pool1_3x3_s2", maxpool3x3_s2(y))
    for name, cin, mid, cout, s, d, proj in BNECKS[:4]:
        y = keep(name, bottleneck(P, name, y, s, d, proj, ep))
    f2 = y
    # low-resolution branch
    y = keep("conv3_1_sub4", resize_bilinear(f2, h // 32, w // 32))
    for name, cin, mid, cout, s, d, proj in BNECKS[4:]:
        y = keep(name, bottleneck(P, name, y, s, d, proj, ep))
    y = keep("conv5_3_sum", pyramid_pooling(y))
    f1 = keep("conv5_4_k1", conv_bn(P, "conv5_4_k1", y))
    # high-resolution branch
    y = keep("conv1_sub1", conv_bn(P, "conv1_sub1", x, stride=2))
    y = keep("conv2_sub1", conv_bn(P, "conv2_sub1", y, stride=2))
    f3 = keep("conv3_sub1", conv_bn(P, "conv3_sub1", y, stride=2))
    # cascade feature fusion + head
    s24, up, pj = cff(P, f1, f2, "conv_sub4", "conv3_1_sub2_proj")
    keep("conv5_4_interp", up)
    keep("conv3_1_sub2_proj", pj)
    keep("sub24_sum", s24)
    s12, up, pj = cff(P, s24, f3, "conv_sub2", "conv3_sub1_proj")
    keep("sub24_sum_interp", up)
    keep("conv3_sub1_proj", pj)
    keep("sub12_sum", s12)
    y = keep("sub12_sum_interp", resize_bilinear(s12, h // 4, w // 4))
    y = eo.conv2d_same(y, P["conv6_cls.kernel"])
    y = keep("conv6_cls", affine_add_relu(y, None, P["conv6_cls.bias"], None, relu=False))
    if not full_logits:
        return y
    return keep("conv6_interp", resize_bilinear(y, h, w))


def score_images(P, x, measure="margin"):
    """-> (mean float64 [N], per-pixel confidence, label uint8, logits) -- active_learning.py:234-263 on ICNet"""
    logits = icnet_forward(P, x)
    mean, conf, label = eo.score_logits(logits, measure)
    return mean, conf, label, logits


def macs_per_image(h, w, c_in=3, classes=19):
    """multiply-accumulates of one forward pass (convolutions only)"""
    # resolution of each conv's OUTPUT relative to the input
    total = 0
    res = {}
    for name, k, cin, cout, s, d in conv_specs(c_in, classes):
        if name in ("conv1_1_3x3_s2", "conv1_2_3x3", "conv1_3_3x3"):
            div = 4
        elif name.startswith("conv2_"):
            div = 8
        elif name.startswith("conv3_1_") and "sub2" not in name:
            div = 16
        elif name.startswith(("conv3_", "conv4_", "conv5_")) and "sub" not in name:
            div = 32
        elif name in ("conv_sub4", "conv3_1_sub2_proj"):
            div = 16
        elif name in ("conv_sub2", "conv3_sub1_proj", "conv3_sub1"):
            div = 8
        elif name == "conv1_sub1":
            div = 2
        elif name == "conv2_sub1":
            div = 4
        elif name == "conv6_cls":
            div = 4
        else:
            raise KeyError(name)
        res[name] = div
        total += (h // div) * (w // div) * k * k * cin * cout
    return total
