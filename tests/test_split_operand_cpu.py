"""Accuracy side of the "beyond the fp32 wall" experiment (the timing side is
tools/mfma_peak.py shapes 516 / 616 in the measurement library, profiles/r04_probe_mfma_peak_and_bf16x3_split.txt).

Every fp32 MFMA-bound kernel of the path is capped by the fp32 issue rate.  The way past it would be to split each fp32
operand into three bf16 terms (x = x1 + x2 + x3 exactly, by truncation) and run the 6 leading cross products on
v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  NOTHING of this ships: the product library, the bench headline and
`dtype: f32` stay exact fp32.  This test prices what such a path would do to the logits: a whole ENet forward on one
256x512 frame, evaluated (a) in float64, (b) in float32 (the shipping arithmetic, torch's summation order), (c) in float32
with every convolution replaced by the bf16x3 emulation -- errors are taken GIVEN EQUAL POOLING WINNERS (a flipped
max-pool winner moves a value by a pixel and is a property of any two evaluations, not an accuracy figure)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_restatement as tr
from semanticsegmentationactivelearning_amd import synthetic as syn
import semanticsegmentationactivelearning_amd as ssal

MASK = torch.tensor(-65536, dtype=torch.int32)  # 0xffff0000


def _trunc_bf16(x):
    """the bf16 value obtained by dropping the low 16 bits of an fp32 value (as fp32)"""
    return (x.contiguous().view(torch.int32) & MASK).view(torch.float32)


def split3(x):
    x1 = _trunc_bf16(x)
    r1 = x - x1
    x2 = _trunc_bf16(r1)
    r2 = r1 - x2
    return x1, x2, _trunc_bf16(r2)


def _split_op(op):
    def f(x, w, **kw):
        xs, ws = split3(x), split3(w)
        # the 6 leading cross products, small terms first; each term is a sum of EXACT bf16 x bf16 products in fp32
        y = op(xs[2], ws[0], **kw)
        for i, j in ((1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
            y = y + op(xs[i], ws[j], **kw)
        return y
    return f


def test_split_is_exact_and_products_fit():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(100000, generator=g) * torch.exp(4 * torch.randn(100000, generator=g))
    x1, x2, x3 = split3(x)
    assert torch.equal((x1.double() + x2.double() + x3.double()).float(), x)  # three truncated terms carry all 24 bits
    # a bf16 x bf16 product has <= 16 significant bits: exact in fp32
    p = (x1[:50000].double() * x2[50000:].double())
    assert torch.equal(p.float().double(), p)


@pytest.fixture(scope="module")
def forwards():
    net = ssal.ENet(19)
    net.build((None, None, None, 3))
    syn.randomize_enet(net, seed=0)
    P = syn.enet_params_dict(net)
    x = syn.synth_frames_f32([7], 256, 512, 3)
    out = {}
    try:
        tr.DTYPE = torch.float64
        ep = {}
        out["f64"] = tr.enet_forward(P, x, ep)
        winners = {"argmax1": ep["argmax1"], "argmax2": ep["argmax2"]}
        tr.DTYPE = torch.float32
        out["f32"] = tr.enet_forward(P, x, None, winners)
        tr.CONV2D, tr.CONV_T2D = _split_op(F.conv2d), _split_op(F.conv_transpose2d)
        out["bf16x3"] = tr.enet_forward(P, x, None, winners)
    finally:
        tr.DTYPE, tr.CONV2D, tr.CONV_T2D = torch.float32, F.conv2d, F.conv_transpose2d
    return out


def test_bf16x3_split_forward_error_next_to_fp32(forwards):
    ref = forwards["f64"].astype(np.float64)
    scale = float(np.abs(ref).max())
    rows = {}
    for name in ("f32", "bf16x3"):
        d = forwards[name].astype(np.float64) - ref
        rows[name] = (float(np.abs(d).max()), float(np.sqrt(np.mean(d * d))))
        print("%-7s logits vs float64: max |d| %.3e  rms %.3e   (|logit| max %.1f)" % (name, rows[name][0], rows[name][1], scale))
    # exact fp32 sits at a few 1e-5 on logits of magnitude ~30; the split path must stay in the same decade (it drops
    # three O(2^-24) cross terms per product) -- comfortably inside the 1e-4 budget of north_star's softmax tolerance
    assert rows["f32"][0] < 1e-3 and rows["bf16x3"][0] < 1e-3
    assert rows["bf16x3"][1] < 4.0 * rows["f32"][1] + 1e-7
    labels = {k: v.argmax(-1) for k, v in forwards.items()}
    flips = int((labels["bf16x3"] != labels["f64"]).sum()), int((labels["f32"] != labels["f64"]).sum())
    print("argmax labels differing from float64: bf16x3 %d, f32 %d of %d pixels" % (flips[0], flips[1], labels["f64"].size))
    assert flips[0] <= flips[1] + 16
