"""TEST INFRASTRUCTURE ONLY -- second, independent CPU restatement ("restatement B")
of the reference scoring path on stock torch-CPU fp32 ops.  Never imported by the product path.

Purpose: (1) cross-check the C oracle (different kernels, different accumulation order, un-folded
batch-norm exactly as TF writes it) -- both must agree to <= 1e-4 on logits; (2) serve as the
multi-threaded CPU baseline timed by ``bench.py`` ("cpu_baseline", kind "port"): it is the fastest
honest CPU form of the reference path available here (the real TensorFlow-1.13 CPU path cannot be
run: TensorFlow is not installed and there is no network).

TF-1.13 semantics encoded (reference call sites in parentheses):
  * conv2d SAME = explicit asymmetric zero pad (total//2 before, rest after) + VALID correlation
    (enet_modules.py:205,538,...,1285)
  * conv2d_transpose 3x3/s2 SAME = conv_transpose2d(stride 2, padding 0) cropped to [:2H,:2W]
    (enet_modules.py:1251-1255,1376-1380)
  * fused_batch_norm(is_training=False): (x-mean)*rsqrt(var+1e-3)*gamma+beta (extra_ops.py:181-184)
  * prelu: relu(x) - alpha*relu(-x) (extra_ops.py:21-26)
  * max_pool_with_argmax: per-image flattened index (y*W+x)*C+c; unpool_2d = scatter
    into zeros (extra_ops.py:82-85)
  * softmax / entropy / margin / confidence / float64 mean (active_learning.py:239-263)
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPSILON = float(np.finfo(np.float32).tiny)


# Hooks of tests/test_split_operand_cpu.py (the accuracy side of the "beyond the fp32 wall" experiment): DTYPE = float64
# turns the restatement into the fp64-accumulated evaluation; CONV2D / CONV_T2D swap the convolution primitive.
DTYPE = torch.float32
CONV2D = F.conv2d
CONV_T2D = F.conv_transpose2d


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DTYPE)


def _same_pad(size, k, stride, dil):
    out = -(-size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - size, 0)
    return total // 2, total - total // 2


def conv2d_same(x, w_hwio, stride=1, dil=1):
    """x: NCHW tensor; w: numpy HWIO"""
    kh, kw = w_hwio.shape[:2]
    pt, pb = _same_pad(x.shape[2], kh, stride, dil)
    pl, pr = _same_pad(x.shape[3], kw, stride, dil)
    x = F.pad(x, (pl, pr, pt, pb))
    w = _t(w_hwio).permute(3, 2, 0, 1).contiguous()  # OIHW
    return CONV2D(x, w, stride=stride, dilation=dil)


def conv2d_transpose_3x3_s2(x, w_hwoi):
    """kernel [3,3,O,I] (TF) -> torch conv_transpose2d weight [I,O,3,3]; crop row/col 2H, 2W."""
    w = _t(w_hwoi).permute(3, 2, 0, 1).contiguous()
    y = CONV_T2D(x, w, stride=2, padding=0)
    return y[:, :, : 2 * x.shape[2], : 2 * x.shape[3]].contiguous()


def batch_norm(x, P, prefix):
    m, v = _t(P[prefix + "mean"]), _t(P[prefix + "variance"])
    g, b = _t(P[prefix + "gamma"]), _t(P[prefix + "beta"])
    sh = (1, -1, 1, 1)
    return (x - m.view(sh)) * torch.rsqrt(v.view(sh) + 1e-3) * g.view(sh) + b.view(sh)


def prelu(x, alpha):
    a = _t(alpha).view(1, -1, 1, 1)
    return F.relu(x) - a * F.relu(-x)


def max_pool_with_argmax(x):
    """-> pooled NCHW, argmax in the reference's per-image NHWC-flattened convention"""
    n, c, h, w = x.shape
    y, idx = F.max_pool2d(x, 2, 2, return_indices=True)  # idx = iy*W + ix per (n,c) plane
    ch = torch.arange(c).view(1, c, 1, 1)
    return y, idx * c + ch


def unpool_2d(x, argmax):
    """scatter into zeros([N,2H,2W,C]) with per-image indices (y*W+x)*C+c"""
    n, c, h, w = x.shape
    out = torch.zeros((n, 2 * h * 2 * w * c), dtype=x.dtype)
    out.scatter_(1, argmax.permute(0, 2, 3, 1).reshape(n, -1), x.permute(0, 2, 3, 1).reshape(n, -1))
    return out.view(n, 2 * h, 2 * w, c).permute(0, 3, 1, 2).contiguous()


def initial(P, nm, x):
    conv = conv2d_same(x, P[nm + ".kernel"], stride=2)
    pool = F.max_pool2d(x, 2, 2)
    return prelu(batch_norm(torch.cat([conv, pool], 1), P, nm + "."), P[nm + ".alpha"])


def bottleneck(P, nm, x, dil=1, asym=False):
    y = prelu(batch_norm(conv2d_same(x, P[nm + ".proj_kernel"]), P, nm + ".proj_"), P[nm + ".proj_alpha"])
    if asym:
        y = conv2d_same(conv2d_same(y, P[nm + ".conv_kernel.0"]), P[nm + ".conv_kernel.1"])
    else:
        y = conv2d_same(y, P[nm + ".conv_kernel"], dil=dil)
    y = prelu(batch_norm(y, P, nm + ".conv_"), P[nm + ".conv_alpha"])
    y = batch_norm(conv2d_same(y, P[nm + ".exp_kernel"]), P, nm + ".exp_")
    return prelu(y + x, P[nm + ".residual_alpha"])


def bottleneck_down(P, nm, x):
    y = prelu(batch_norm(conv2d_same(x, P[nm + ".proj_kernel"], stride=2), P, nm + ".proj_"), P[nm + ".proj_alpha"])
    y = prelu(batch_norm(conv2d_same(y, P[nm + ".conv_kernel"]), P, nm + ".conv_"), P[nm + ".conv_alpha"])
    y = batch_norm(conv2d_same(y, P[nm + ".exp_kernel"]), P, nm + ".exp_")
    pool, argmax = max_pool_with_argmax(x)
    pool = F.pad(pool, (0, 0, 0, 0, 0, y.shape[1] - pool.shape[1]))  # zero channels at the end
    return prelu(y + pool, P[nm + ".residual_alpha"]), argmax


def bottleneck_up(P, nm, x, argmax):
    y = prelu(batch_norm(conv2d_same(x, P[nm + ".proj_kernel"]), P, nm + ".proj_"), P[nm + ".proj_alpha"])
    y = prelu(batch_norm(conv2d_transpose_3x3_s2(y, P[nm + ".conv_kernel"]), P, nm + ".conv_"), P[nm + ".conv_alpha"])
    y = batch_norm(conv2d_same(y, P[nm + ".exp_kernel"]), P, nm + ".exp_")
    res = unpool_2d(conv2d_same(x, P[nm + ".res_kernel"]), argmax)
    return prelu(y + res, P[nm + ".residual_alpha"])


_STAGE23 = [(1, 1, False), (2, 2, False), (3, 1, True), (4, 4, False), (5, 1, False), (6, 8, False),
            (7, 1, True), (8, 16, False)]


@torch.no_grad()
def enet_forward(P, x_nhwc, endpoints=None, pooling_indices=None):
    """numpy NHWC in -> numpy NHWC logits.

    pooling_indices = {"argmax1": [N,H/4,W/4,64] int64, "argmax2": [N,H/8,W/8,128]} (reference convention) makes the two
    unpool layers scatter to THOSE positions instead of this restatement's own argmax.  Two fp32 evaluations of the same
    network round differently, so a pooling window whose two largest values are within an ulp or two can elect a different
    winner; the unpool layer then writes the (practically equal) value to a different pixel and the logits around it
    differ by O(1) -- a property of max-pool-with-argmax + unpool under ANY change of summation order (it would show
    against TensorFlow's own kernels just the same), not an error of either side.  Comparing 'given the same winners'
    separates that effect from real numerical disagreement."""
    ep = endpoints if endpoints is not None else {}

    def rec(name, t):
        ep[name] = t.permute(0, 2, 3, 1).contiguous().numpy()
        return t

    x = _t(x_nhwc).permute(0, 3, 1, 2).contiguous()
    y = rec("Initial", initial(P, "Initial", x))
    y, a1 = bottleneck_down(P, "Bottleneck1_0", y)
    rec("Bottleneck1_0", y)
    ep["argmax1"] = a1.permute(0, 2, 3, 1).contiguous().numpy()
    if pooling_indices is not None:
        a1 = torch.from_numpy(np.ascontiguousarray(pooling_indices["argmax1"]).astype(np.int64)).permute(0, 3, 1, 2)
    for i in range(1, 5):
        y = rec("Bottleneck1_%d" % i, bottleneck(P, "Bottleneck1_%d" % i, y))
    y, a2 = bottleneck_down(P, "Bottleneck2_0", y)
    rec("Bottleneck2_0", y)
    ep["argmax2"] = a2.permute(0, 2, 3, 1).contiguous().numpy()
    if pooling_indices is not None:
        a2 = torch.from_numpy(np.ascontiguousarray(pooling_indices["argmax2"]).astype(np.int64)).permute(0, 3, 1, 2)
    for stage in (2, 3):
        for i, dil, asym in _STAGE23:
            nm = "Bottleneck%d_%d" % (stage, i)
            y = rec(nm, bottleneck(P, nm, y, dil=dil, asym=asym))
    y = rec("Bottleneck4_0", bottleneck_up(P, "Bottleneck4_0", y, a2))
    y = rec("Bottleneck4_1", bottleneck(P, "Bottleneck4_1", y))
    y = rec("Bottleneck4_2", bottleneck(P, "Bottleneck4_2", y))
    y = rec("Bottleneck5_0", bottleneck_up(P, "Bottleneck5_0", y, a1))
    y = rec("Bottleneck5_1", bottleneck(P, "Bottleneck5_1", y))
    y = rec("Final", conv2d_transpose_3x3_s2(y, P["Final.kernel"]))
    return ep["Final"]


@torch.no_grad()
def score_logits(logits_nhwc, measure="entropy"):
    """active_learning.py:239-263 literally (fp32 per pixel, float64 mean)."""
    lg = _t(logits_nhwc)
    k = lg.shape[-1]
    prob = torch.softmax(lg, dim=-1)
    if measure == "entropy":
        ent = -(prob * torch.log(prob + EPSILON)).sum(-1)
        conf = 1.0 - ent / math.log(np.float32(k))
    elif measure == "margin":
        v, _ = torch.topk(prob, 2, dim=-1)
        conf = v[..., 0] - v[..., 1]
    elif measure == "confidence":
        conf = prob.max(-1).values
    else:
        raise NotImplementedError("Uncertainty function not implemented.")
    mean = conf.double().mean(dim=(1, 2))
    label = lg.argmax(-1).to(torch.uint8)
    return mean.numpy(), conf.numpy(), label.numpy()


@torch.no_grad()
def score_images(P, x_nhwc, measure="entropy"):
    logits = enet_forward(P, x_nhwc)
    mean, conf, label = score_logits(logits, measure)
    return mean, conf, label, logits


# ====================================================================================================
# ICNet (ICNET_SPEC.md) -- independent restatement on stock torch-CPU ops.  There is no reference behaviour for
# this network (models/icnet/icnet.py:1-7 is an empty class): "parity unpinned AND undefined".
# ====================================================================================================
def resize_bilinear_legacy(x, oh, ow):
    """tf.image.resize_bilinear, TF-1.13 defaults (align_corners=False, src = dst * in/out, no half-pixel offset)
    on an NCHW tensor -- torch's own interpolate() uses half-pixel centres, so the gather is written out."""
    n, c, h, w = x.shape
    fy = torch.arange(oh, dtype=torch.float32) * (float(h) / float(oh))
    fx = torch.arange(ow, dtype=torch.float32) * (float(w) / float(ow))
    y0, x0 = fy.floor().long(), fx.floor().long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    ly = (fy - y0.float()).view(1, 1, oh, 1)
    lx = (fx - x0.float()).view(1, 1, 1, ow)
    top = x[:, :, y0][:, :, :, x0] + (x[:, :, y0][:, :, :, x1] - x[:, :, y0][:, :, :, x0]) * lx
    bot = x[:, :, y1][:, :, :, x0] + (x[:, :, y1][:, :, :, x1] - x[:, :, y1][:, :, :, x0]) * lx
    return top + (bot - top) * ly


def max_pool_3x3_s2_same(x):
    h, w = x.shape[2:]
    pt, pb = _same_pad(h, 3, 2, 1)
    pl, pr = _same_pad(w, 3, 2, 1)
    return F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=float("-inf")), 3, 2)


def _ic_conv_bn(P, nm, x, stride=1, dil=1, relu=True, res=None):
    y = batch_norm(conv2d_same(x, P[nm + ".kernel"], stride=stride, dil=dil), P, nm + ".")
    if res is not None:
        y = y + res
    return F.relu(y) if relu else y


def _ic_bottleneck(P, nm, x, stride, dil, proj):
    sc = _ic_conv_bn(P, nm + "_1x1_proj", x, stride=stride, relu=False) if proj else x
    y = _ic_conv_bn(P, nm + "_1x1_reduce", x, stride=stride)
    y = _ic_conv_bn(P, nm + "_3x3", y, dil=dil)
    return _ic_conv_bn(P, nm + "_1x1_increase", y, relu=True, res=sc)


_IC_BNECKS = ([("conv2_1", 1, 1, True), ("conv2_2", 1, 1, False), ("conv2_3", 1, 1, False), ("conv3_1", 2, 1, True)]
              + [("conv3_%d" % i, 1, 1, False) for i in (2, 3, 4)] + [("conv4_1", 1, 2, True)]
              + [("conv4_%d" % i, 1, 2, False) for i in (2, 3, 4, 5, 6)] + [("conv5_1", 1, 4, True)]
              + [("conv5_%d" % i, 1, 4, False) for i in (2, 3)])


def icnet_forward(P, x_nhwc, endpoints=None, full_logits=True):
    """ICNET_SPEC.md sections 1-4 -> logits NHWC numpy"""
    with torch.no_grad():
        x = _t(x_nhwc).permute(0, 3, 1, 2).contiguous()
        n, _, h, w = x.shape
        ep = endpoints if endpoints is not None else {}
        y = resize_bilinear_legacy(x, h // 2, w // 2)
        y = _ic_conv_bn(P, "conv1_1_3x3_s2", y, stride=2)
        y = _ic_conv_bn(P, "conv1_2_3x3", y)
        y = _ic_conv_bn(P, "conv1_3_3x3", y)
        y = max_pool_3x3_s2_same(y)
        for nm, s, d, proj in _IC_BNECKS[:4]:
            y = _ic_bottleneck(P, nm, y, s, d, proj)
        f2 = y
        y = resize_bilinear_legacy(f2, h // 32, w // 32)
        for nm, s, d, proj in _IC_BNECKS[4:]:
            y = _ic_bottleneck(P, nm, y, s, d, proj)
        ep["conv5_3"] = y
        hh, ww = y.shape[2:]
        acc = y
        for b in (1, 2, 3, 6):
            acc = acc + resize_bilinear_legacy(F.adaptive_avg_pool2d(y, b), hh, ww)
        ep["conv5_3_sum"] = acc
        f1 = _ic_conv_bn(P, "conv5_4_k1", acc)
        y = _ic_conv_bn(P, "conv1_sub1", x, stride=2)
        y = _ic_conv_bn(P, "conv2_sub1", y, stride=2)
        f3 = _ic_conv_bn(P, "conv3_sub1", y, stride=2)
        up = resize_bilinear_legacy(f1, 2 * f1.shape[2], 2 * f1.shape[3])
        s24 = _ic_conv_bn(P, "conv_sub4", up, dil=2, relu=True, res=_ic_conv_bn(P, "conv3_1_sub2_proj", f2, relu=False))
        ep["sub24_sum"] = s24
        up = resize_bilinear_legacy(s24, 2 * s24.shape[2], 2 * s24.shape[3])
        s12 = _ic_conv_bn(P, "conv_sub2", up, dil=2, relu=True, res=_ic_conv_bn(P, "conv3_sub1_proj", f3, relu=False))
        ep["sub12_sum"] = s12
        y = resize_bilinear_legacy(s12, h // 4, w // 4)
        y = conv2d_same(y, P["conv6_cls.kernel"]) + _t(P["conv6_cls.bias"]).view(1, -1, 1, 1)
        ep["conv6_cls"] = y
        for k in list(ep):
            if isinstance(ep[k], torch.Tensor):
                ep[k] = ep[k].permute(0, 2, 3, 1).contiguous().numpy()
        if full_logits:
            y = resize_bilinear_legacy(y, h, w)
        return y.permute(0, 2, 3, 1).contiguous().numpy()


def icnet_score_images(P, x_nhwc, measure="margin"):
    logits = icnet_forward(P, x_nhwc)
    mean, conf, label = score_logits(logits, measure)
    return mean, conf, label, logits
