"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded
inputs.  Integer / index results (pool argmax, class argmax, top-k ids) must be bit-exact; fp32
activations and logits are compared bit-for-bit against the C restatement (same accumulation
order) AND within 1e-4 of the independent torch-CPU restatement; softmax / entropy / margin within
1e-4 (north_star tolerance, written at each assert).
"""
import os

import numpy as np
import pytest
import torch

import semanticsegmentationactivelearning_amd as ssal
from helpers import frames, pool_score_table, report_diff
from oracle import enet_oracle as orc
from oracle import torch_restatement as tr
from semanticsegmentationactivelearning_amd import _lib, active_learning as al, synthetic as syn
from semanticsegmentationactivelearning_amd.models.util import extra_ops as xops

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4  # north_star: softmax/entropy within 1e-4 fp32


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    _lib.lib()  # the HIP extension must be the thing that runs
    yield
    torch.cuda.synchronize()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---- synthetic frames: device generator == host twin ---------------------------------------------
@pytest.mark.parametrize("h,w,c,first,count", [(64, 128, 3, 0, 3), (32, 32, 4, 2970, 2), (8, 8, 1, 5, 1)])
def test_synth_frames_device_equals_host(h, w, c, first, count):
    got = syn.synth_frames_device(first, count, h, w, c).cpu().numpy()
    want = frames(range(first, first + count), h, w, c)
    report_diff("frames", got, want)


def test_uint8_frames_same_bits_as_float32_frames(enet_c3k19, enet_c4k6):
    """the decoded uint8 frame goes straight into the Initial block (converted on the fly like
    tf.image.convert_image_dtype, input.py:289-290): logits, labels and scores must equal the float32 path"""
    u8 = syn.synth_frames_device(70, 3, 64, 96, 3, dtype=torch.uint8)
    host = np.stack([syn.synth_frame_u8(f, 64, 96, 3) for f in (70, 71, 72)])
    assert u8.dtype == torch.uint8 and (u8.cpu().numpy() == host).all()
    for (net, P), c in ((enet_c3k19, 3), (enet_c4k6, 4)):
        xu = syn.synth_frames_device(70, 3, 64, 96, c, dtype=torch.uint8)
        xf = syn.synth_frames_device(70, 3, 64, 96, c)
        assert torch.equal(net(xu, training=False), net(xf, training=False))
        su, eu = net.score(xu, "entropy", return_label=True, return_confidence=True)
        sf, ef = net.score(xf, "entropy", return_label=True, return_confidence=True)
        assert torch.equal(su, sf) and torch.equal(eu["label"], ef["label"]) and torch.equal(eu["confidence"], ef["confidence"])
        # numpy uint8 input takes the same route
        assert torch.equal(net.score(xu.cpu().numpy(), "margin"), net.score(xf, "margin"))


# ---- stand-alone operators -----------------------------------------------------------------------
@pytest.mark.parametrize("kh,kw,stride,dil,h,w,cin,cout", [
    (1, 1, 1, 1, 16, 24, 64, 16), (1, 1, 1, 1, 9, 7, 16, 64), (1, 1, 1, 1, 8, 8, 32, 128), (1, 1, 1, 1, 8, 8, 128, 32),
    (3, 3, 1, 1, 16, 20, 16, 16), (3, 3, 1, 2, 16, 20, 32, 32), (3, 3, 1, 4, 9, 11, 32, 32),
    (3, 3, 1, 8, 16, 24, 32, 32), (3, 3, 1, 16, 16, 40, 32, 32), (5, 1, 1, 1, 12, 10, 32, 32),
    (1, 5, 1, 1, 12, 10, 32, 32), (2, 2, 2, 1, 16, 24, 16, 8), (2, 2, 2, 1, 8, 8, 64, 32),
    (3, 3, 1, 1, 10, 6, 4, 4), (3, 3, 1, 1, 6, 5, 8, 8), (3, 3, 2, 1, 9, 7, 4, 8),
])
def test_conv2d_same_bit_exact(kh, kw, stride, dil, h, w, cin, cout):
    rng = np.random.default_rng(10)
    x = rng.normal(size=(2, h, w, cin)).astype(np.float32)
    k = (rng.normal(size=(kh, kw, cin, cout)) * 0.2).astype(np.float32)
    want = orc.conv2d_same(x, k, stride, dil)
    y = torch.empty(want.shape, dtype=torch.float32, device="cuda")
    xd, kd = dev(x), dev(k)  # keep the device tensors alive across the call
    _lib.check(_lib.lib().ssal_conv2d_same(_lib.dev_ptr(xd), 2, h, w, cin, _lib.dev_ptr(kd), kh, kw, cout,
                                           stride, dil, _lib.dev_ptr(y), _lib.stream_ptr()))
    report_diff("conv2d", y.cpu().numpy(), want)


@pytest.mark.parametrize("h,w,cin,cout", [(8, 12, 32, 16), (5, 7, 16, 8), (4, 4, 16, 16), (1, 1, 4, 4)])
def test_conv2d_transpose_bit_exact(h, w, cin, cout):
    rng = np.random.default_rng(11)
    x = rng.normal(size=(2, h, w, cin)).astype(np.float32)
    k = (rng.normal(size=(3, 3, cout, cin)) * 0.2).astype(np.float32)
    want = orc.conv2d_transpose_3x3_s2(x, k)
    y = torch.empty(want.shape, dtype=torch.float32, device="cuda")
    xd, kd = dev(x), dev(k)
    _lib.check(_lib.lib().ssal_conv2d_transpose_3x3_s2(_lib.dev_ptr(xd), 2, h, w, cin, _lib.dev_ptr(kd),
                                                       cout, _lib.dev_ptr(y), _lib.stream_ptr()))
    report_diff("conv2d_transpose", y.cpu().numpy(), want)


def test_conv_argument_errors():
    x = torch.zeros((1, 4, 4, 3), device="cuda")
    k = torch.zeros((3, 3, 3, 8), device="cuda")
    y = torch.zeros((1, 4, 4, 8), device="cuda")
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().ssal_conv2d_same(_lib.dev_ptr(x), 1, 4, 4, 3, _lib.dev_ptr(k), 3, 3, 8, 1, 1,
                                               _lib.dev_ptr(y), _lib.stream_ptr()))


@pytest.mark.parametrize("include_batch", [False, True])
def test_max_pool_with_argmax_and_unpool(include_batch):
    rng = np.random.default_rng(12)
    x = rng.uniform(size=(3, 16, 24, 5)).astype(np.float32)
    x[0, :2, :2, 0] = 0.5  # a tie: first element of the window must win
    want_y, want_i = orc.maxpool2x2_argmax(x, include_batch=include_batch)
    y, idx = xops.max_pool_with_argmax(dev(x), include_batch_in_index=include_batch)
    report_diff("pool", y.cpu().numpy(), want_y)
    report_diff("argmax", idx.cpu().numpy(), want_i)
    up = xops.unpool_2d(y, idx, idx_has_batch=include_batch)
    report_diff("unpool", up.cpu().numpy(), orc.unpool2d(want_y, want_i, idx_has_batch=include_batch))
    # reference invariant (models/util/test_xops.py:6-21): pool(unpool(pool(x))) == pool(x), error exactly 0
    y2, _ = xops.max_pool_with_argmax(up)
    assert float((y - y2).abs().sum()) == 0.0


def test_prelu_and_batch_norm_ops():
    rng = np.random.default_rng(13)
    x = rng.normal(size=(2, 6, 5, 16)).astype(np.float32)
    a = rng.uniform(-0.5, 0.5, 16).astype(np.float32)
    report_diff("prelu", xops.prelu(dev(x), dev(a)).cpu().numpy(), orc.affine_prelu(x, None, None, a))
    m, v = rng.normal(0, 0.1, 16).astype(np.float32), rng.uniform(0.5, 1.5, 16).astype(np.float32)
    g, b = rng.uniform(0.8, 1.2, 16).astype(np.float32), rng.normal(0, 0.1, 16).astype(np.float32)
    s, t = orc.bn_fold(m, v, g, b)
    out, um, uv = xops.batch_norm(dev(x), dev(m), dev(v), dev(g), dev(b), training=False)
    assert um is None and uv is None
    report_diff("batch_norm", out.cpu().numpy(), orc.affine_prelu(x, s, t, None))
    with pytest.raises(NotImplementedError):
        xops.batch_norm(dev(x), dev(m), dev(v), dev(g), dev(b), training=True)


@pytest.mark.parametrize("c,oh,ow", [(3, 13, 10), (8, 13, 10), (8, 2, 3), (64, 10, 14)])
def test_resize_bilinear_tf113_legacy_mapping(c, oh, ow):
    """tf.image.resize_bilinear defaults in TF 1.13: align_corners=False, src = dst * in/out; channel counts that take the scalar
    kernel (3) and the channel-quad kernel (8, 64), up- and down-scaling"""
    rng = np.random.default_rng(14)
    x = rng.normal(size=(2, 5, 7, c)).astype(np.float32)
    y = torch.empty((2, oh, ow, c), dtype=torch.float32, device="cuda")
    xd = dev(x)
    _lib.check(_lib.lib().ssal_resize_bilinear(_lib.dev_ptr(xd), 2, 5, 7, c, oh, ow, _lib.dev_ptr(y), _lib.stream_ptr()))
    hs, ws = np.float32(5) / np.float32(oh), np.float32(7) / np.float32(ow)
    want = np.empty((2, oh, ow, c), np.float32)
    for oy in range(oh):
        fy = np.float32(oy) * hs
        y0 = int(np.floor(fy)); y1 = min(y0 + 1, 4); ly = np.float32(fy - np.float32(y0))
        for ox in range(ow):
            fx = np.float32(ox) * ws
            x0 = int(np.floor(fx)); x1 = min(x0 + 1, 6); lx = np.float32(fx - np.float32(x0))
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            want[:, oy, ox] = top + (bot - top) * ly
    report_diff("resize_bilinear", y.cpu().numpy(), want, exact=False, atol=1e-6)


# ---- score kernel on materialised logits ---------------------------------------------------------
@pytest.mark.parametrize("measure", ["entropy", "margin", "confidence"])
@pytest.mark.parametrize("n,h,w,k", [(2, 16, 32, 19), (1, 5, 7, 6), (3, 24, 40, 2), (1, 8, 8, 32), (2, 3, 3, 19)])
def test_score_logits_kernel(measure, n, h, w, k):
    rng = np.random.default_rng(15)
    lg = (rng.normal(size=(n, h, w, k)) * 4).astype(np.float32)
    lg[0, 0, 0, :] = 1.25  # all classes tie: label 0, margin 0, entropy conf 0
    want_mean, want_conf, want_label = orc.score_logits(lg, measure)
    thr = float(np.median(want_conf))
    scores, extra = al.score_logits(dev(lg), measure, threshold=thr, return_label=True, return_mask=True,
                                    return_confidence=True)
    report_diff("label (bit-exact argmax)", extra["label"].cpu().numpy(), want_label)
    conf = extra["confidence"].cpu().numpy()
    report_diff("confidence", conf, want_conf, exact=False, atol=TOL)  # 1e-4 tolerance (north_star)
    assert np.abs(conf - want_conf).max() < 5e-6  # what the kernel actually achieves
    report_diff("mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    mask = extra["mask"].cpu().numpy()
    sure = np.abs(want_conf - thr) > 1e-5
    assert ((mask == (want_conf >= thr))[sure]).all()
    assert (mask == (conf >= np.float32(thr))).all()  # active_learning.py:265-269 on the kernel's own conf


def test_score_logits_errors():
    lg = torch.zeros((1, 4, 4, 19), device="cuda")
    with pytest.raises(NotImplementedError):
        al.score_logits(lg, "bald")
    with pytest.raises(ValueError):
        al.score_logits(torch.zeros((1, 4, 4, 40), device="cuda"), "entropy")


# ---- single layers through ssal_enet_run_layer ---------------------------------------------------
def _layer_cases():
    return [("Initial", 16, 24), ("Bottleneck1_0", 16, 24), ("Bottleneck1_1", 12, 20), ("Bottleneck2_0", 8, 12),
            ("Bottleneck2_1", 9, 11), ("Bottleneck2_2", 9, 11), ("Bottleneck2_3", 9, 11), ("Bottleneck2_4", 12, 9),
            ("Bottleneck2_6", 18, 20), ("Bottleneck3_7", 6, 13), ("Bottleneck3_8", 34, 36), ("Bottleneck4_0", 6, 8),
            ("Bottleneck4_2", 8, 8), ("Bottleneck5_0", 6, 10), ("Bottleneck5_1", 10, 12), ("Final", 8, 12)]


@pytest.mark.parametrize("name,h,w", _layer_cases())
def test_single_layer_bit_exact(enet_c3k19, name, h, w):
    net, P = enet_c3k19
    layer = getattr(net, name)
    kind = type(layer).__name__
    cin = {"Initial": 3, "Final": 16}.get(name)
    if cin is None:
        cin = layer.proj_kernel.shape[2]
    rng = np.random.default_rng(16)
    x = rng.normal(size=(2, h, w, cin)).astype(np.float32)
    dil = layer.dilation_rate[0] if kind == "Bottleneck" else 1
    if kind == "Initial":
        report_diff(name, layer(dev(x), training=False).cpu().numpy(), orc.initial(P, name, x))
    elif kind == "Bottleneck":
        want = orc.bottleneck(P, name, x, dil=dil, asym=layer.asymmetric)
        report_diff(name, layer(dev(x), training=False).cpu().numpy(), want)
    elif kind == "BottleneckDownsample":
        want, want_arg = orc.bottleneck_down(P, name, x)
        y, arg = layer(dev(x), training=False)
        report_diff(name, y.cpu().numpy(), want)
        report_diff(name + " argmax (bit-exact)", arg.cpu().numpy(), want_arg)
    elif kind == "BottleneckUpsample":
        cout = layer.output_channels
        pre = rng.normal(size=(2, 2 * h, 2 * w, cout)).astype(np.float32)
        _, arg = orc.maxpool2x2_argmax(pre)
        want = orc.bottleneck_up(P, name, x, arg)
        report_diff(name, layer(dev(x), dev(arg), training=False).cpu().numpy(), want)
    else:
        report_diff(name, layer(dev(x)).cpu().numpy(), orc.final(P, name, x))


# ---- MFMA-fused bottleneck kernels: hardware assumptions + equality with the generic family --------
def test_permlane32_swap_lane_semantics():
    """swap32(a, b) must give a = [a.lo | b.lo], b = [a.hi | b.hi] (ssal_bottleneck_mfma.hip relies on it)"""
    out = torch.zeros(256, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().ssal_debug_probe(_lib.dev_ptr(out), _lib.stream_ptr()))
    got = out.cpu().numpy()
    lane = np.arange(64, dtype=np.float32)
    a, b = lane, 100 + lane
    assert (got[:64] == np.concatenate([a[:32], b[:32]])).all(), got[:64]
    assert (got[64:128] == np.concatenate([a[32:], b[32:]])).all(), got[64:128]
    # swap16(a, b): a = [a.q0 b.q0 a.q2 b.q2], b = [a.q1 b.q1 a.q3 b.q3]  (q = 16-lane quarters)
    qa, qb = a.reshape(4, 16), b.reshape(4, 16)
    assert (got[128:192] == np.concatenate([qa[0], qb[0], qa[2], qb[2]])).all(), got[128:192]
    assert (got[192:256] == np.concatenate([qa[1], qb[1], qa[3], qb[3]])).all(), got[192:256]


@pytest.mark.parametrize("name,n,h,w", [
    ("Bottleneck2_1", 2, 16, 32), ("Bottleneck2_1", 1, 9, 11), ("Bottleneck2_1", 1, 8, 40), ("Bottleneck2_2", 2, 16, 32),
    ("Bottleneck2_2", 1, 17, 35), ("Bottleneck2_4", 1, 32, 64), ("Bottleneck2_4", 2, 12, 9), ("Bottleneck2_6", 1, 18, 20),
    ("Bottleneck2_6", 1, 32, 64), ("Bottleneck2_8", 1, 34, 36), ("Bottleneck3_8", 2, 32, 64), ("Bottleneck3_5", 3, 24, 72),
    ("Bottleneck3_8", 1, 16, 16), ("Bottleneck2_1", 1, 1, 1),
    ("Bottleneck1_1", 2, 16, 32), ("Bottleneck1_3", 1, 9, 11), ("Bottleneck1_4", 1, 24, 72), ("Bottleneck4_1", 2, 32, 64),
    ("Bottleneck4_2", 1, 1, 1), ("Bottleneck1_2", 1, 8, 40),
    ("Bottleneck2_3", 2, 16, 32), ("Bottleneck2_7", 1, 9, 11), ("Bottleneck3_3", 1, 24, 72), ("Bottleneck3_7", 2, 32, 64),
    ("Bottleneck2_3", 1, 1, 1), ("Bottleneck3_7", 1, 8, 40), ("Bottleneck2_7", 1, 5, 3),
    ("Bottleneck5_1", 2, 16, 32), ("Bottleneck5_1", 1, 9, 11), ("Bottleneck5_1", 1, 24, 72), ("Bottleneck5_1", 1, 1, 1),
])
def test_mfma_bottleneck_equals_generic_and_oracle(enet_c3k19, name, n, h, w):
    net, P = enet_c3k19
    layer = getattr(net, name)
    x = np.random.default_rng(18).normal(size=(n, h, w, layer.output_channels)).astype(np.float32)
    want = orc.bottleneck(P, name, x, dil=layer.dilation_rate[0], asym=layer.asymmetric)
    xd = dev(x)
    try:
        _lib.set_kernel_family(True)
        got_mfma = layer(xd, training=False).cpu().numpy()
        _lib.set_kernel_family(False)
        got_gen = layer(xd, training=False).cpu().numpy()
    finally:
        _lib.set_kernel_family(True)
    report_diff(name + " generic vs oracle", got_gen, want)
    report_diff(name + " MFMA vs oracle (bit-exact)", got_mfma, want)


@pytest.mark.parametrize("name", ["Bottleneck2_1", "Bottleneck2_2", "Bottleneck2_3", "Bottleneck3_8"])
def test_bottleneck_tile_shapes_bit_identical(enet_c3k19, name):
    """8x32 and 8x16 tiles, XCD-aware and plain tile order of the 128-channel bottleneck kernels must produce
    the same bits, at a size with several tiles per CU"""
    net, P = enet_c3k19
    layer = getattr(net, name)
    x = np.random.default_rng(23).normal(size=(3, 128, 256, 128)).astype(np.float32)
    xd = dev(x)
    try:
        ref = layer(xd, training=False)
        _lib.set_knob("bnk_tw", 16)
        got = layer(xd, training=False)
        assert torch.equal(got, ref), "8x16 tiles differ from 8x32 tiles"
        _lib.set_knob("bnk_tw", 0)
        _lib.set_knob("bnk_xcd", 0)
        got = layer(xd, training=False)
        assert torch.equal(got, ref), "plain tile order differs from the XCD-aware order"
        _lib.set_knob("bnk_xcd", 1)
        for o4 in (0, 1):  # default 2: k_bottleneck_o4 only where the phase sub-image is at most 16 pixels wide (3_8 here)
            _lib.set_knob("bnk_o4", o4)
            got = layer(xd, training=False)
            assert torch.equal(got, ref), "bnk_o4=%d (k_bottleneck_o4: 8x16 tiles, four workgroups per CU) differs" % o4
        # asymmetric block: the default k_bottleneck_mfma_asym16x (8x16 tiles, the (5,1) result written over the projected
        # rows, three workgroups per CU) against the 8x32 / two-halves kernel of rounds 1-4
        assert _lib.get_knobs()["asym_tw16"] == 1
        _lib.set_knob("asym_tw16", 0)
        got = layer(xd, training=False)
        assert torch.equal(got, ref), "asym_tw16=0 (k_bottleneck_mfma_asym<32>) differs from the default kernel"
        _lib.set_knob("asym_tw16", 1)
        # regular block: the default expansion epilogue of k_bottleneck_mfma<32> (D[co][pixel], 16-byte residual loads, whole-row
        # stores through quad_transpose4) against the same without the transpose and the D[pixel][co] epilogue of rounds 1-4
        assert _lib.get_knobs()["bnk_qepi"] == 2
        for qepi in (1, 0):
            _lib.set_knob("bnk_qepi", qepi)
            got = layer(xd, training=False)
            assert torch.equal(got, ref), "bnk_qepi=%d differs from the default epilogue" % qepi
    finally:
        _lib.set_knob("bnk_tw", 0)
        _lib.set_knob("bnk_xcd", 1)
        _lib.set_knob("bnk_o4", 2)
        _lib.set_knob("asym_tw16", 1)
        _lib.set_knob("bnk_qepi", 2)
    want = orc.bottleneck(P, name, x[1:2], dil=layer.dilation_rate[0], asym=layer.asymmetric)
    report_diff(name + " [128,256] vs oracle (bit-exact)", ref[1:2].cpu().numpy(), want)


@pytest.mark.parametrize("name,n,h,w", [("Bottleneck4_0", 2, 16, 32), ("Bottleneck4_0", 1, 9, 11), ("Bottleneck4_0", 1, 8, 40),
                                        ("Bottleneck4_0", 1, 1, 1), ("Bottleneck5_0", 2, 16, 32), ("Bottleneck5_0", 1, 7, 5)])
def test_upsample_layer_both_unpool_forms_and_families(enet_c3k19, name, n, h, w):
    """pooling-derived indices take the window-code (gather) form, arbitrary indices the scatter form"""
    net, P = enet_c3k19
    layer = getattr(net, name)
    cin, cout = layer.proj_kernel.shape[2], layer.output_channels
    rng = np.random.default_rng(19)
    x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
    _, arg_pool = orc.maxpool2x2_argmax(rng.normal(size=(n, 2 * h, 2 * w, cout)).astype(np.float32))
    # arbitrary (but unique per image) indices: a random permutation of the output positions
    arg_any = np.stack([rng.permutation(4 * h * w * cout)[: h * w * cout].reshape(h, w, cout) for _ in range(n)])
    xd = dev(x)
    for arg in (arg_pool, arg_any):
        want = orc.bottleneck_up(P, name, x, arg)
        try:
            for fam in (True, False):
                _lib.set_kernel_family(fam)
                got = layer(xd, dev(arg), training=False).cpu().numpy()
                report_diff("%s family=%s" % (name, "mfma" if fam else "generic"), got, want)
        finally:
            _lib.set_kernel_family(True)


@pytest.mark.parametrize("name,n,h,w", [("Bottleneck2_0", 2, 16, 32), ("Bottleneck2_0", 1, 18, 22), ("Bottleneck2_0", 1, 16, 80),
                                        ("Bottleneck2_0", 1, 2, 2), ("Bottleneck1_0", 2, 16, 32), ("Bottleneck1_0", 1, 6, 10)])
def test_downsample_layer_both_families(enet_c3k19, name, n, h, w):
    net, P = enet_c3k19
    layer = getattr(net, name)
    cin = layer.proj_kernel.shape[2]
    rng = np.random.default_rng(20)
    x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
    x[0, :2, :2, :] = 0.25  # an all-equal pooling window: the first element must win
    want, want_arg = orc.bottleneck_down(P, name, x)
    xd = dev(x)
    try:
        for fam in (True, False):
            _lib.set_kernel_family(fam)
            got, arg = layer(xd, training=False)
            tag = "%s family=%s" % (name, "mfma" if fam else "generic")
            report_diff(tag, got.cpu().numpy(), want)
            report_diff(tag + " argmax (bit-exact)", arg.cpu().numpy(), want_arg)
    finally:
        _lib.set_kernel_family(True)


# ---- whole network -------------------------------------------------------------------------------
def _check_forward(net, P, x, tag):
    ep = {}
    want = orc.enet_forward(P, x, ep)
    got = net(dev(x), training=False).cpu().numpy()
    final, b5_1, b4_2, b3_8 = net.endpoint_outputs[-1]
    report_diff(tag + " bottleneck3_8", b3_8.cpu().numpy(), ep["Bottleneck3_8"])
    report_diff(tag + " bottleneck4_2", b4_2.cpu().numpy(), ep["Bottleneck4_2"])
    report_diff(tag + " bottleneck5_1", b5_1.cpu().numpy(), ep["Bottleneck5_1"])
    report_diff(tag + " logits vs C oracle (bit-exact)", got, want)
    wep = {}
    wb = tr.enet_forward(P, x, wep)
    # two fp32 evaluations may elect different winners in a pooling window whose two largest values are an ulp apart
    # (1 window of 524 288 on the 512x1024 frame of test_forward_reference_conf_frame_sizes); the unpool layer then moves
    # the value by a pixel and the logits around it differ by O(1).  Such windows must be rare, and given the SAME
    # winners the two restatements must agree within the tolerance everywhere.
    flips = sum(int((wep[a] != ep[a]).sum()) for a in ("argmax1", "argmax2"))
    if flips:
        assert flips <= max(1, int(1e-5 * (ep["argmax1"].size + ep["argmax2"].size))), flips
        wb = tr.enet_forward(P, x, pooling_indices={a: ep[a] for a in ("argmax1", "argmax2")})
    report_diff(tag + " logits vs torch restatement", got, wb, exact=False, atol=TOL)
    return got, want


def test_forward_c1_256x512(enet_c3k19):
    """BASELINE config C1: 4 synthetic 256x512x3 frames, K=19: forward + entropy + top-1"""
    net, P = enet_c3k19
    x = frames([0, 1, 2, 3], 256, 512, 3)
    got, want = _check_forward(net, P, x, "C1")
    want_mean, want_conf, want_label = orc.score_logits(want, "entropy")
    scores, extra = net.score(dev(x), "entropy", return_label=True, return_confidence=True)
    report_diff("C1 pseudo_label (bit-exact argmax)", extra["label"].cpu().numpy(), want_label)
    report_diff("C1 confidence", extra["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL)
    report_diff("C1 mean confidence", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    low, _ = al.finish_ranking(np.arange(4), scores.cpu().numpy(), 4, np.arange(4), 1)
    want_low, _ = orc.rank_lowest(want_mean, np.arange(4), 1)
    assert low.tolist() == want_low.tolist()


def test_forward_matches_golden_fixture(enet_c3k19):
    net, P = enet_c3k19
    g = np.load(os.path.join(GOLDEN, "enet_c3k19_64x128.npz"))
    x = frames(list(g["frame_ids"]), 64, 128, 3)
    got = net(dev(x), training=False).cpu().numpy()
    report_diff("logits vs golden", got, g["logits"])
    for m in ("entropy", "margin", "confidence"):
        scores, extra = net.score(dev(x), m, return_label=True, return_confidence=True)
        report_diff(m + " label vs golden", extra["label"].cpu().numpy(), g["label"])
        report_diff(m + " conf vs golden", extra["confidence"].cpu().numpy()[0], g["conf_" + m], exact=False, atol=TOL)
        report_diff(m + " mean vs golden", scores.cpu().numpy(), g["mean_" + m], exact=False, atol=1e-6)


def test_forward_c5_rgb_nir_6_classes(enet_c4k6):
    """BASELINE config C5 shape family: 4-channel input (RGB+NIR), 6 classes"""
    net, P = enet_c4k6
    g = np.load(os.path.join(GOLDEN, "enet_c4k6_64x64.npz"))
    x = frames(list(g["frame_ids"]), 64, 64, 4)
    got, _ = _check_forward(net, P, x, "C5")
    report_diff("C5 logits vs golden", got, g["logits"])
    scores = net.score(dev(x), "entropy")
    report_diff("C5 mean", scores.cpu().numpy(), g["mean_entropy"], exact=False, atol=1e-6)


@pytest.mark.parametrize("n,h,w", [(1, 8, 8), (3, 24, 40), (1, 136, 72), (5, 16, 8)])
def test_forward_ragged_shapes(enet_c3k19, n, h, w):
    net, P = enet_c3k19
    x = frames(range(20, 20 + n), h, w, 3)
    _check_forward(net, P, x, "%dx%dx%d" % (n, h, w))


@pytest.mark.parametrize("classes,c_in", [(2, 3), (3, 1), (7, 3), (20, 4), (31, 3), (32, 3)])
def test_forward_and_score_other_class_counts(classes, c_in):
    """the Final conv runs two classes per packed FMA: odd / even / minimal / maximal class counts (and 1- and
    4-channel inputs) must still give bit-exact logits and labels and the oracle's scores"""
    from helpers import make_model
    net, P = make_model(classes, c_in, seed=5)
    x = frames([60, 61], 16, 24, c_in)
    got, _ = _check_forward(net, P, x, "K=%d" % classes)
    for m in ("entropy", "margin"):
        scores, extra = net.score(dev(x), m, return_label=True, return_confidence=True)
        want_mean, want_conf, want_label, _ = orc.score_images(P, x, m)
        report_diff("K=%d %s label" % (classes, m), extra["label"].cpu().numpy(), want_label)
        report_diff("K=%d %s conf" % (classes, m), extra["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL)
        report_diff("K=%d %s mean" % (classes, m), scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)


def test_fused_score_equals_unfused_path(enet_c3k19):
    """the fused Final+score kernel and (logits -> stand-alone score kernel) agree"""
    net, _ = enet_c3k19
    x = dev(frames([7, 8], 64, 128, 3))
    logits = net(x, training=False)
    for m in ("entropy", "margin", "confidence"):
        a, ea = net.score(x, m, threshold=0.3, return_label=True, return_mask=True, return_confidence=True)
        b, eb = al.score_logits(logits, m, threshold=0.3, return_label=True, return_mask=True, return_confidence=True)
        assert torch.equal(ea["label"], eb["label"]) and torch.equal(ea["mask"], eb["mask"])
        assert torch.equal(ea["confidence"], eb["confidence"])
        assert float((a - b).abs().max()) < 1e-12


def test_score_is_bitwise_reproducible(enet_c3k19):
    net, _ = enet_c3k19
    x = dev(frames([3, 4, 5], 64, 128, 3))
    a = net.score(x, "entropy").clone()
    b = net.score(x, "entropy").clone()
    assert torch.equal(a, b)
    # batch-size independence: images are independent at inference
    c = torch.cat([net.score(x[i:i + 1], "entropy") for i in range(3)])
    assert torch.equal(a, c)


def test_repeated_scoring_is_bitwise_stable_at_bench_tile_counts(enet_c3k19):
    """soak: 25 back-to-back passes over the same 2 x 512 x 1024 batch (thousands of workgroups per launch, both tile
    widths of the 128-channel kernels through the dilation-16 layers) give identical scores, labels and confidences"""
    net, _ = enet_c3k19
    x = syn.synth_frames_device(300, 2, 512, 1024, 3)
    s0, e0 = net.score(x, "entropy", return_label=True, return_confidence=True)
    s0, l0, c0 = s0.clone(), e0["label"].clone(), e0["confidence"].clone()
    for _ in range(24):
        s, e = net.score(x, "entropy", return_label=True, return_confidence=True)
        assert torch.equal(s, s0) and torch.equal(e["label"], l0) and torch.equal(e["confidence"], c0)


def test_weight_update_is_picked_up(enet_c3k19):
    net, P = enet_c3k19
    x = frames([1], 32, 32, 3)
    before = net(dev(x), training=False).cpu().numpy()
    old = net.Final.kernel.numpy().copy()
    net.Final.kernel.assign(old * np.float32(0.5))  # active_learning.py:461-462 re-initialises Final.kernel
    try:
        P2 = dict(P)
        P2["Final.kernel"] = net.Final.kernel.numpy()
        after = net(dev(x), training=False).cpu().numpy()
        report_diff("after assign", after, orc.enet_forward(P2, x))
        assert not np.array_equal(before, after)
    finally:
        net.Final.kernel.assign(old)


def test_input_errors(enet_c3k19):
    net, _ = enet_c3k19
    with pytest.raises(ValueError, match="divisible by 8"):
        net(torch.zeros((1, 20, 16, 3), device="cuda"), training=False)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 16, 16, 4), device="cuda"), training=False)
    with pytest.raises(NotImplementedError):
        net(torch.zeros((1, 16, 16, 3), device="cuda"), training=True)
    with pytest.raises(NotImplementedError):
        net.score(torch.zeros((1, 16, 16, 3), device="cuda"), measure="bald")


def test_rank_confidence_topk_ids_match_oracle(enet_c3k19):
    """rank_confidence (active_learning.py:682-715): identical top-k example ids as the oracle"""
    net, P = enet_c3k19
    num, bs, k = 13, 4, 3
    order = np.random.default_rng(17).permutation(num)  # the reference feeds a shuffled order
    unlabelled = np.array([0, 1, 2, 4, 5, 7, 8, 9, 11, 12])

    def batches():
        for i in range(0, num, bs):  # last batch is partial (input.py:193-194, no drop_remainder)
            ids = order[i:i + bs]
            yield frames(ids, 64, 64, 3), ids

    low, uc = al.rank_confidence(net, batches(), num, unlabelled, k, measure="entropy")
    want_scores = np.concatenate([orc.score_images(P, frames([i], 64, 64, 3))[0] for i in range(num)])
    want_low, want_uc = orc.rank_lowest(want_scores, unlabelled, k)
    gap = np.sort(want_uc)[k] - np.sort(want_uc)[k - 1]
    assert gap > 1e-4, "fixture must have a decision margin at the k-th boundary (gap=%g)" % gap
    assert set(low.tolist()) == set(want_low.tolist())
    assert uc.dtype == np.float32
    report_diff("unlabelled_confidence", uc, want_uc, exact=False, atol=1e-6)


def test_full_resolution_image_bit_exact(enet_c3k19):
    """BASELINE C2 shape: one 1024x2048x3 frame, checked against the C oracle (~20 s of CPU)"""
    net, P = enet_c3k19
    x = frames([100], 1024, 2048, 3)
    ep = {}
    want = orc.enet_forward(P, x, ep)
    xd = dev(x)
    scores, extra = net.score(xd, "entropy", return_label=True)
    got = net(xd, training=False).cpu().numpy()
    report_diff("1024x2048 logits (bit-exact)", got, want)
    want_mean, _, want_label = orc.score_logits(want, "entropy")
    report_diff("1024x2048 label", extra["label"].cpu().numpy(), want_label)
    report_diff("1024x2048 mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    # the frame just checked against the oracle is entry 100 of the table bench.py's score_digest is compared with
    table = pool_score_table("enet", 3, 19, 1024, 2048, "entropy", 0)
    assert scores.cpu().numpy()[0] == table[100], "HIP score of frame 100 != committed pool_scores.npz entry"
    # ... and the SHIPPING ranking-pass form (no outputs: Initial + 1_0 in one launch, 5_1 inside Final + score) of the same
    # frame gives the same bits as the unfused tail that was just compared with the oracle
    assert net.score(xd, "entropy").cpu().numpy()[0] == table[100], "fused-ends score of frame 100 != oracle-checked score"


def test_full_resolution_c5_rgb_nir_frame_bit_exact(enet_c4k6):
    """BASELINE configs[4] at its stated size: one 1024x2048x4 (RGB + NIR) frame, 6 classes (reference
    datasets/freiburg.py:47, conf/freiburg_forest.json), entropy -- logits / labels bit-exact against the C oracle,
    per-image mean <= 1e-6 (~20 s of CPU)"""
    net, P = enet_c4k6
    x = frames([100], 1024, 2048, 4)
    want = orc.enet_forward(P, x, {})
    xd = dev(x)
    scores, extra = net.score(xd, "entropy", return_label=True)
    got = net(xd, training=False).cpu().numpy()
    assert got.shape == (1, 1024, 2048, 6)
    report_diff("C5 1024x2048x4 logits (bit-exact)", got, want)
    want_mean, _, want_label = orc.score_logits(want, "entropy")
    report_diff("C5 1024x2048x4 label", extra["label"].cpu().numpy(), want_label)
    report_diff("C5 1024x2048x4 mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    table = pool_score_table("enet", 4, 6, 1024, 2048, "entropy", 1)
    assert scores.cpu().numpy()[0] == table[100], "HIP score of C5 frame 100 != committed pool_scores.npz entry"
    assert net.score(xd, "entropy").cpu().numpy()[0] == table[100], "fused-ends score of C5 frame 100 != oracle-checked score"


def test_pool_score_table_entries_reproduce_in_other_batch_compositions(enet_c3k19):
    """bench.py compares the SHA-256 of the scores it timed with the same frames of tests/golden/pool_scores.npz: entries of
    that table must be reproduced bit for bit whatever batch a frame is scored in (batches of 3, 1 and 8 here; the table
    was produced in batches of 8 starting at multiples of 8)"""
    net, _ = enet_c3k19
    table = pool_score_table("enet", 3, 19, 1024, 2048, "entropy", 0)
    assert table.shape == (2975,) and np.isfinite(table).all()
    for first, count in ((5, 3), (2974, 1), (1480, 8)):
        got = net.score(syn.synth_frames_device(first, count, 1024, 2048, 3), "entropy").cpu().numpy()
        assert np.array_equal(got, table[first:first + count]), (first, got, table[first:first + count])
    srt = np.sort(table.astype(np.float32))
    assert srt[128] - srt[127] > 0, "top-128 boundary of the bench pool must not be a tie"


def test_rank_confidence_from_tfrecords(enet_c3k19, tmp_path):
    """end-to-end front-end -> GPU: one PNG example per .tfrecord, decoded by tensortools.InputStage (the
    reference's train/rank path with the (labelled, index) side channels), scored on the GPU, ranked"""
    from test_input_cpu import write_pool
    from semanticsegmentationactivelearning_amd.tensortools import InputStage, NumpyCapsule
    net, P = enet_c3k19
    num, k = 9, 2
    files = write_pool(str(tmp_path), num, 64, 64, with_label=False)
    cap = NumpyCapsule(shuffle=True, seed=5)
    cap.filenames, cap.indices = files, np.arange(num)
    cap.labelled = np.zeros(num, dtype=bool)
    stage = InputStage(input_shape=[64, 64], seed=6)
    stage.add_dataset_from_placeholders("train", cap.filenames, cap.labelled, cap.indices, batch_size=4)
    stage.init_iterator("train", None, cap.feed_dict)  # eval-style decode: centre crop == identity, no flip

    def batches():
        for image, label, mask, labelled, index in stage:
            yield image, index

    unlabelled = np.arange(num)
    low, uc = al.rank_confidence(net, batches(), num, unlabelled, k, measure="margin")
    want_scores = np.concatenate([orc.score_images(P, frames([i], 64, 64, 3), "margin")[0] for i in range(num)])
    want_low, want_uc = orc.rank_lowest(want_scores, unlabelled, k)
    assert set(low.tolist()) == set(want_low.tolist())
    report_diff("unlabelled_confidence", uc, want_uc, exact=False, atol=1e-6)


def test_rank_confidence_from_tfrecords_pinned_uint8_prefetch(enet_c3k19, tmp_path):
    """the production feed: TFRecords -> InputStage(uint8 frames in page-locked batches) -> side-stream copy ->
    GPU conversion + scoring; must rank exactly like the plain float32 path"""
    from test_input_cpu import write_pool
    from semanticsegmentationactivelearning_amd.tensortools import InputStage
    net, P = enet_c3k19
    num = 11
    write_pool(str(tmp_path), num, 64, 64, with_label=False)
    results = []
    for kwargs, prefetch in (({}, 0), ({"image_dtype": np.uint8, "pin_memory": True, "pin_buffers": 4}, 2)):
        stage = InputStage(input_shape=[64, 64], **kwargs)
        stage.add_dataset("val", str(tmp_path), batch_size=3)
        stage.init_iterator("val")
        pos = [0]

        def batches():
            for image, label, mask in stage:
                n = len(image)
                yield image, np.arange(pos[0], pos[0] + n)
                pos[0] += n
        results.append(al.rank_confidence(net, batches(), num, np.arange(num), 3, prefetch=prefetch))
    (low_a, uc_a), (low_b, uc_b) = results
    assert set(low_a.tolist()) == set(low_b.tolist()) and (uc_a == uc_b).all()


def test_pinned_ring_slot_reuse_with_rewrapped_batches(tmp_path):
    """pin_buffers=2 with prefetch=2 forces every ring slot to be rewritten while earlier copies may still be
    in flight; the batches reach prefetch_to_device sliced and re-wrapped (numpy round trip -> torch.from_numpy), i.e.
    WITHOUT any Python attribute of the tensor InputStage yielded.  The copy-done events are keyed by the address of the
    page-locked memory (tensortools.input.copy_issued), so every frame that arrives must equal its source."""
    from test_input_cpu import write_pool
    from semanticsegmentationactivelearning_amd.tensortools import InputStage, input as tin
    num, h, w = 14, 96, 160
    write_pool(str(tmp_path), num, h, w, with_label=False)
    ref = InputStage(input_shape=[h, w], image_dtype=np.uint8)
    ref.add_dataset("val", str(tmp_path), batch_size=3)
    ref.init_iterator("val")
    want = np.concatenate([np.asarray(b[0]) for b in ref])
    stage = InputStage(input_shape=[h, w], image_dtype=np.uint8, pin_memory=True, pin_buffers=2)
    stage.add_dataset("val", str(tmp_path), batch_size=3)
    stage.init_iterator("val")
    seen_slots = []

    def batches():
        pos = 0
        for image, label, mask in stage:
            assert image.is_pinned()
            alias = torch.from_numpy(image.numpy()[:, :, :, :])  # new tensor object, same page-locked memory
            ev = torch.cuda.Event()
            ev.record()
            assert tin.copy_issued(alias[1:], ev) is True  # a slice of the re-wrapped batch resolves to its slot
            assert tin.copy_issued(np.zeros(4, dtype=np.uint8), ev) is False  # foreign memory: nothing to protect
            seen_slots.append(alias.data_ptr())
            yield alias, np.arange(pos, pos + len(alias))
            pos += len(alias)
    got = [x.clone() for x, _ in al.prefetch_to_device(batches(), depth=2)]
    torch.cuda.synchronize()
    got = torch.cat(got).cpu().numpy()
    assert len(set(seen_slots)) == 2 and len(seen_slots) == 5  # 5 batches through 2 slots: each slot reused
    assert np.array_equal(got, want)
    assert all(len(v) == 0 or all(e.query() for e in v) for v in stage._pin_events.values())


def test_rank_confidence_prefetch_and_uint8_host_batches(enet_c3k19):
    """host batches (uint8 decoded frames) copied ahead on a side stream give the same ranking as device batches"""
    net, P = enet_c3k19
    num, bs = 10, 4
    host = [np.stack([syn.synth_frame_u8(f, 64, 64, 3) for f in range(b, min(b + bs, num))]) for b in range(0, num, bs)]

    def host_batches():
        for k, x in enumerate(host):
            yield x, np.arange(k * bs, k * bs + len(x))

    def dev_batches():
        for k, x in enumerate(host):
            yield dev(syn.u8_to_f32(x)), np.arange(k * bs, k * bs + len(x))
    low_a, uc_a = al.rank_confidence(net, dev_batches(), num, np.arange(num), 3)
    for pf in (1, 2, 5):
        low_b, uc_b = al.rank_confidence(net, host_batches(), num, np.arange(num), 3, prefetch=pf)
        assert set(low_a.tolist()) == set(low_b.tolist()) and (uc_a == uc_b).all()


# ---- loss forward value, inference path ------------------------------------
@pytest.mark.parametrize("weight,ls,k", [(0.0, 0.0, 19), (0.0, 0.05, 19), (1.5, 0.05, 19), (1.02, 0.1, 6)])
def test_masked_softmax_cross_entropy_forward(weight, ls, k):
    from semanticsegmentationactivelearning_amd.tensortools import losses
    rng = np.random.default_rng(21)
    lg = (rng.normal(size=(3, 12, 20, k)) * 3).astype(np.float32)
    lab = rng.integers(0, k, size=(3, 12, 20)).astype(np.uint8)
    mask = (rng.uniform(size=(3, 12, 20)) > 0.3).astype(np.float32)
    want = orc.masked_softmax_cross_entropy(lab, lg, mask, k, weight, ls)
    got = float(losses.masked_softmax_cross_entropy(dev(lab), dev(lg), dev(mask), k, weight, ls))
    assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (got, want)
    # literal float64 numpy statement of tensortools/losses.py:27-73
    x = lg.astype(np.float64)
    lse = np.log(np.exp(x - x.max(-1, keepdims=True)).sum(-1)) + x.max(-1)
    oh = np.full(lg.shape, ls / (k - 1.0)); np.put_along_axis(oh, lab[..., None].astype(np.int64), 1.0 - ls, -1)
    ce = (oh * (lse[..., None] - x)).sum(-1) * mask
    if weight > 1.0:
        p = (np.exp(x - lse[..., None]) * oh).sum(-1)
        ce = ce / np.log(weight + (1.718281828459045 - weight) * p)
    ref = ce.sum(0).sum() / mask.sum()
    assert abs(got - ref) <= 2e-5 * max(1.0, abs(ref)), (got, ref)
    l2 = losses.L2_regularization([np.ones((2, 2), np.float32), 2 * np.ones(3, np.float32)], 0.5)
    assert abs(l2 - 0.5 / 2 * (4 / 2 + 12 / 2)) < 1e-12


def test_multiscale_cross_entropy_forward_on_endpoint_outputs(enet_c3k19):
    """tensortools/losses.py:76-157 on ENet.endpoint_outputs (final logits + the 1/2, 1/4, 1/8 features):
    1x1 heads, nearest-neighbour resized labels / mask, sum of the per-scale losses"""
    from semanticsegmentationactivelearning_amd.tensortools import losses
    net, P = enet_c3k19
    rng = np.random.default_rng(24)
    x = frames([40, 41], 64, 96, 3)
    lab = rng.integers(0, 19, size=(2, 64, 96)).astype(np.uint8)
    mask = (rng.uniform(size=(2, 64, 96)) > 0.25).astype(np.float32)
    net(dev(x), training=False)
    outs = net.endpoint_outputs[-1]
    assert [tuple(o.shape) for o in outs] == [(2, 64, 96, 19), (2, 32, 48, 16), (2, 16, 24, 64), (2, 8, 12, 128)]
    got, heads = losses.multiscale_masked_softmax_cross_entropy(dev(lab), outs, dev(mask), 19, weight=1.5,
                                                                label_smoothing=0.05, seed=3)
    assert [k.shape for k in heads] == [(1, 1, 16, 19), (1, 1, 64, 19), (1, 1, 128, 19)]
    want = orc.multiscale_masked_softmax_cross_entropy(lab, [o.cpu().numpy() for o in outs], mask, 19, heads,
                                                       weight=1.5, label_smoothing=0.05)
    assert abs(float(got) - want) <= 1e-5 * max(1.0, abs(want)), (float(got), want)
    # explicit head kernels are honoured; normalize=True fails as the reference's `len(loss)` does
    got2, _ = losses.multiscale_masked_softmax_cross_entropy(dev(lab), outs, dev(mask), 19, kernels=heads)
    want2 = orc.multiscale_masked_softmax_cross_entropy(lab, [o.cpu().numpy() for o in outs], mask, 19, heads)
    assert abs(float(got2) - want2) <= 1e-5 * max(1.0, abs(want2))
    with pytest.raises(TypeError):
        losses.multiscale_masked_softmax_cross_entropy(dev(lab), outs, dev(mask), 19, normalize=True)
    # nearest-neighbour index rule: src = floor(dst * in / out)
    t = torch.arange(2 * 6 * 9, device="cuda", dtype=torch.float32).reshape(2, 6, 9)
    r = losses.resize_nearest_neighbor(t, (3, 4)).cpu().numpy()
    assert (r == t.cpu().numpy()[:, [0, 2, 4]][:, :, [0, 2, 4, 6]]).all()


def test_inference_path_labels_embedding_and_png(enet_c3k19, tmp_path):
    from semanticsegmentationactivelearning_amd import inference as inf
    from PIL import Image
    net, P = enet_c3k19
    x = frames([30, 31], 64, 64, 3)
    want_logits = orc.enet_forward(P, x)
    pred = inf.predict_labels(net, dev(x))
    report_diff("trainId map (bit-exact argmax)", pred.cpu().numpy(), want_logits.argmax(-1).astype(np.uint8))
    emb = np.zeros(256, np.uint8); emb[:19] = [7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33]
    ids = inf.reverse_embedding(pred, emb).cpu().numpy()
    assert (ids == emb[want_logits.argmax(-1)]).all()
    # resized logits (reference inference.py:96-99): compare where the decision margin is not razor thin
    up = inf.resize_bilinear(dev(want_logits), (96, 80)).cpu().numpy()
    pred_up = inf.predict_labels(net, dev(x), size=(96, 80)).cpu().numpy()
    srt = np.sort(up, -1)
    sure = (srt[..., -1] - srt[..., -2]) > 1e-4
    assert (pred_up[sure] == up.argmax(-1)[sure]).all() and sure.mean() > 0.99
    paths = inf.run_inference(net, [(x, [b"a", "b"])], str(tmp_path / "out"), embedding_reversed=emb)
    assert [os.path.basename(p) for p in paths] == ["a.png", "b.png"]
    assert (np.asarray(Image.open(paths[1])) == ids[1]).all()


def test_forward_c5_medium_rgb_nir(enet_c4k6):
    """BASELINE config C5 shape family at 256x512: 4-channel input (RGB+NIR), 6 classes, entropy"""
    net, P = enet_c4k6
    x = frames([40, 41], 256, 512, 4)
    got, want = _check_forward(net, P, x, "C5-256x512")
    want_mean, _, want_label = orc.score_logits(want, "entropy")
    scores, extra = net.score(dev(x), "entropy", return_label=True)
    report_diff("C5 label", extra["label"].cpu().numpy(), want_label)
    report_diff("C5 mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)


def test_forward_reference_conf_frame_sizes(enet_c3k19, enet_c4k6):
    """The frame sizes the reference's own parameter files configure (conf/freiburg_forest.json:33-34: 432x648, whose
    stage-2/3 maps are 54x81 -- odd width, partial tiles in every kernel; conf/enet_cityscapes_*.json:33-34: 512x1024):
    forward + entropy score + label, RGB+NIR / 6 classes for Freiburg (BASELINE config C5), 3 channels / 19 classes for
    Cityscapes."""
    for (net, P), h, w, c, tag in ((enet_c4k6, 432, 648, 4, "freiburg-432x648"), (enet_c3k19, 512, 1024, 3, "cityscapes-512x1024")):
        x = frames([7], h, w, c)
        got, want = _check_forward(net, P, x, tag)
        want_mean, want_conf, want_label = orc.score_logits(want, "entropy")
        scores, extra = net.score(dev(x), "entropy", return_label=True, return_confidence=True)
        report_diff(tag + " label", extra["label"].cpu().numpy(), want_label)
        report_diff(tag + " conf", extra["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL)
        report_diff(tag + " mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)


# ---- round 2: parity hardening -------------------------------------------------------------------
def test_whole_network_pooling_indices_match_golden_and_oracle(enet_c3k19, enet_c4k6):
    """the max-pooling indices ENet.call hands from the downsample to the upsample blocks (enet.py:331,338), exported
    from the whole-network run in the reference's int64 form, bit-exact against the golden fixtures and the oracle"""
    for (net, P), fx, hw, c in ((enet_c3k19, "enet_c3k19_64x128.npz", (64, 128), 3),
                                (enet_c4k6, "enet_c4k6_64x64.npz", (64, 64), 4)):
        g = np.load(os.path.join(GOLDEN, fx))
        x = frames(list(g["frame_ids"]), hw[0], hw[1], c)
        net(dev(x), training=False)
        a1, a2 = net.pooling_argmax()
        assert a1.dtype == torch.int64 and a2.dtype == torch.int64
        report_diff(fx + " argmax1", a1.cpu().numpy(), g["argmax1"])
        report_diff(fx + " argmax2", a2.cpu().numpy(), g["argmax2"])
        net.score(dev(x), "entropy")  # the score path saves the same indices
        b1, b2 = net.pooling_argmax()
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
    net, P = enet_c3k19
    x = frames([11, 12], 72, 40, 3)
    ep = {}
    orc.enet_forward(P, x, ep)
    net(dev(x), training=False)
    a1, a2 = net.pooling_argmax()
    report_diff("argmax1 vs oracle", a1.cpu().numpy(), ep["argmax1"])
    report_diff("argmax2 vs oracle", a2.cpu().numpy(), ep["argmax2"])


@pytest.mark.parametrize("measure", ["entropy", "margin", "confidence"])
def test_score_edge_vectors(measure):
    """logit gaps beyond exp underflow (p == 0), exact two-way / K-way ties, K = 2, and a threshold exactly equal to
    a pixel's confidence"""
    K = 19
    lg = np.zeros((1, 4, 8, K), np.float32)
    lg[0, 0, 0] = -300.0; lg[0, 0, 0, 7] = 50.0          # one-hot: every other p underflows to exactly 0
    lg[0, 0, 1] = -120.0; lg[0, 0, 1, 3] = 0.0; lg[0, 0, 1, 9] = 0.0   # two-way tie, rest underflow
    lg[0, 0, 2] = 2.5                                     # K-way tie
    lg[0, 0, 3, :] = np.linspace(-150, 10, K)             # mixed: half of the classes underflow
    lg[0, 0, 4, 0] = 88.0; lg[0, 0, 4, 1] = -88.0         # gap 176 with finite exp on one side only
    lg[0, 0, 5, 5] = 1.0; lg[0, 0, 5, 6] = 1.0; lg[0, 0, 5, 7] = 1.0   # three-way tie above a floor
    lg[0, 1] = (np.random.default_rng(3).normal(size=(8, K)) * 30).astype(np.float32)
    want_mean, want_conf, want_label = orc.score_logits(lg, measure)
    assert np.isfinite(want_conf).all()
    scores, extra = al.score_logits(dev(lg), measure, return_label=True, return_confidence=True)
    conf = extra["confidence"].cpu().numpy()
    report_diff("label", extra["label"].cpu().numpy(), want_label)
    report_diff("confidence", conf, want_conf, exact=False, atol=TOL)  # north_star: 1e-4
    report_diff("mean", scores.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    # exact values where the mathematics is exact
    if measure == "margin":
        assert conf[0, 0, 0] == 1.0 and conf[0, 0, 1] == 0.0 and conf[0, 0, 2] == 0.0 and conf[0, 0, 5] == 0.0
    if measure == "confidence":
        assert conf[0, 0, 0] == 1.0 and conf[0, 0, 1] == 0.5 and abs(conf[0, 0, 2] - 1.0 / K) < 1e-7
    if measure == "entropy":
        assert conf[0, 0, 0] == 1.0 and abs(conf[0, 0, 2]) < 1e-6
    assert extra["label"][0, 0, 1].item() == 3 and extra["label"][0, 0, 2].item() == 0  # first maximum wins
    # threshold == a pixel's own confidence: `conf < threshold ? 0 : 1` keeps it (active_learning.py:265-269)
    thr = float(conf[0, 1, 2])
    _, e2 = al.score_logits(dev(lg), measure, threshold=thr, return_mask=True, return_confidence=True)
    mask = e2["mask"].cpu().numpy()
    assert mask[0, 1, 2] == 1
    assert (mask == (conf >= np.float32(thr))).all()
    # K = 2
    l2 = np.array([[[[0.0, 0.0], [200.0, -200.0], [-1.0, 1.0], [3.0, 3.0000002]]]], np.float32)
    wm, wc, wl = orc.score_logits(l2, measure)
    s2, x2 = al.score_logits(dev(l2), measure, return_label=True, return_confidence=True)
    report_diff("K=2 label", x2["label"].cpu().numpy(), wl)
    report_diff("K=2 conf", x2["confidence"].cpu().numpy(), wc, exact=False, atol=TOL)


def test_score_nonfinite_policy():
    """-inf logits behave like exp underflow (p = 0, finite confidence, equal to the oracle's); NaN or +inf logits
    give a NaN confidence for that pixel (like tf.nn.softmax), the image mean is NaN and ranking puts it last"""
    K = 6
    rng = np.random.default_rng(8)
    lg = (rng.normal(size=(2, 4, 4, K)) * 3).astype(np.float32)
    lg[0, 1, 1, 2] = -np.inf
    lg[0, 2, 2, :] = -np.inf
    lg[0, 2, 2, 4] = 0.5
    for m in ("entropy", "margin", "confidence"):
        want_mean, want_conf, want_label = orc.score_logits(lg, m)
        assert np.isfinite(want_conf).all()
        s, e = al.score_logits(dev(lg), m, return_label=True, return_confidence=True)
        report_diff(m + " conf with -inf logits", e["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL)
        report_diff(m + " label", e["label"].cpu().numpy(), want_label)
        assert e["confidence"][0, 2, 2].item() == 1.0
    bad = lg.copy()
    bad[1, 0, 0, 1] = np.nan
    bad[1, 3, 3, 0] = np.inf
    for m in ("entropy", "margin", "confidence"):
        s, e = al.score_logits(dev(bad), m, return_confidence=True)
        c = e["confidence"].cpu().numpy()
        assert np.isnan(c[1, 0, 0]) and np.isnan(c[1, 3, 3])
        assert np.isfinite(c[0]).all() and np.isfinite(np.delete(c[1].ravel(), [0, 15])).all()
        sc = s.cpu().numpy()
        assert np.isfinite(sc[0]) and np.isnan(sc[1])
        low, _ = al.finish_ranking(np.arange(2), sc, 2, np.arange(2), 1)
        assert low.tolist() == [0]  # NaN never ranks as "least confident"


def test_spatial_dropout_op():
    """xops.spatial_dropout (extra_ops.py:137-151): whole (image, channel) planes are either zero or scaled by
    1/(1-rate); exact against oracle/dropout_oracle.py (plain-Python restatement of the seeded draw, no product code) and
    against literal mask bits; deterministic; keep ratio ~ 1-rate"""
    from oracle import dropout_oracle as dorc
    # literal keep bits of (n=2, c=8, rate=0.5, seed=77): pins the draw independently of both implementations
    lit = np.array([[0, 0, 1, 0, 1, 1, 1, 0], [1, 1, 0, 0, 1, 1, 0, 1]], dtype=np.float32)
    assert np.array_equal(dorc.keep_mask(2, 8, 0.5, seed=77), lit)
    ones = np.ones((2, 3, 5, 8), dtype=np.float32)
    assert np.array_equal(xops.spatial_dropout(dev(ones), 0.5, seed=77).cpu().numpy(), 2.0 * ones * lit[:, None, None, :])
    rng = np.random.default_rng(2)
    for (n, h, w, c), rate in (((3, 6, 10, 64), 0.1), ((2, 5, 7, 19), 0.5), ((1, 4, 4, 128), 0.01)):
        x = (rng.normal(size=(n, h, w, c)) + 3.0).astype(np.float32)
        y = xops.spatial_dropout(dev(x), rate, seed=77).cpu().numpy()
        keep = dorc.keep_mask(n, c, rate, seed=77)
        want = dorc.spatial_dropout(x, rate, seed=77)
        report_diff("spatial_dropout", y, want)
        per_plane = (y != 0).reshape(n, h * w, c)
        assert (per_plane.all(axis=1) | (~per_plane).any(axis=1) == True).all()
        assert ((per_plane.all(axis=1)) == (keep == 1)).all() and ((~per_plane).all(axis=1) == (keep == 0)).all()
        y2 = xops.spatial_dropout(dev(x), rate, seed=77).cpu().numpy()
        assert np.array_equal(y, y2)
        y3 = xops.spatial_dropout(dev(x), rate, seed=78).cpu().numpy()
        assert rate < 0.05 or not np.array_equal(y, y3)
    x64 = np.ones((64, 1, 1, 128), dtype=np.float32)
    kept = xops.spatial_dropout(dev(x64), 0.3, seed=5).cpu().numpy() != 0
    assert abs(kept.mean() - 0.7) < 0.02 and np.array_equal(kept[:, 0, 0, :], dorc.keep_mask(64, 128, 0.3, seed=5) == 1)
    assert np.array_equal(xops.spatial_dropout(dev(x), 0.0).cpu().numpy(), x)
    with pytest.raises(ValueError):
        xops.spatial_dropout(dev(x), 1.0)


@pytest.mark.parametrize("classes,c_in,n,h,w", [(19, 3, 3, 8, 8), (19, 3, 2, 24, 40), (6, 4, 1, 136, 72), (2, 1, 5, 16, 8),
                                                (32, 3, 2, 40, 24), (7, 3, 1, 264, 520), (13, 3, 2, 56, 24), (20, 3, 1, 40, 72)])
def test_fused_ends_equal_the_per_layer_launches(classes, c_in, n, h, w):
    """round 3: the ranking pass (score only) runs Initial + Bottleneck1_0 as one launch (k_initial_down16) and evaluates
    Bottleneck5_1 inside the Final + score kernel; a score call that also returns labels takes the per-layer launches for
    5_1 / Final, and `fuse_ends = 0` switches both fusions off.  Every combination must give the same bits, on tiles
    that are cut by the image border on all four sides (odd tile counts, 1 / 3 / 4 input channels, 2 .. 32 classes)."""
    from helpers import make_model
    net, P = make_model(classes, c_in, seed=3)
    x = syn.synth_frames_device(11, n, h, w, c_in)
    want = None
    try:
        for fuse in (3, 1, 2, 0):
            _lib.set_knob("fuse_ends", fuse)
            for measure in ("entropy", "margin", "confidence"):
                a = net.score(x, measure).cpu().numpy()
                b, extra = net.score(x, measure, return_label=True)
                assert np.array_equal(a, b.cpu().numpy()), (fuse, measure)
                if fuse == 3 and measure == "entropy":
                    want = {"label": extra["label"].cpu().numpy()}
                want.setdefault(measure, a)
                assert np.array_equal(a, want[measure]), "fuse_ends=%d changes the %s score" % (fuse, measure)
                assert np.array_equal(extra["label"].cpu().numpy(), want["label"])
    finally:
        _lib.set_knob("fuse_ends", 3)
    ref = orc.score_images(P, syn.synth_frames_f32(np.arange(11, 11 + n), h, w, c_in), "entropy")[0]
    report_diff("fused ends vs oracle", want["entropy"], ref, exact=False, atol=1e-6)


def test_image_group_streams_are_bit_identical(enet_c3k19):
    """the network runs as `img_groups` image chains on library-owned side streams (default: 2 chains over Initial .. Final +
    score): logits, labels and scores must not depend on the grouping (1 = caller's stream only, 2, 4; uneven groups for a
    batch of 3) nor on the span of layers that is grouped, and back-to-back calls on one workspace must not race"""
    net, _ = enet_c3k19
    x = syn.synth_frames_device(40, 4, 128, 256, 3)
    x3 = syn.synth_frames_device(40, 3, 128, 256, 3)
    ref = None
    try:
        for g, span in ((1, 4), (2, 4), (4, 4), (2, 0), (2, 2), (3, 3)):
            _lib.set_knob("img_groups", g)
            _lib.set_knob("img_span", span)
            outs = []
            for _ in range(3):
                s, e = net.score(x, "entropy", return_label=True)
                outs.append((s.cpu().numpy(), e["label"].cpu().numpy(), net(x, training=False).cpu().numpy(),
                             net.score(x3, "entropy").cpu().numpy()))
            for o in outs[1:]:
                assert all(np.array_equal(a, b) for a, b in zip(o, outs[0]))
            if ref is None:
                ref = outs[0]
            assert all(np.array_equal(a, b) for a, b in zip(outs[0], ref)), "img_groups=%d span=%d changes the result" % (g, span)
    finally:
        _lib.set_knob("img_groups", 2)
        _lib.set_knob("img_span", 4)
    assert _lib.get_knobs()["defaults"] == 1


def test_two_host_threads_share_one_handle(enet_c3k19):
    """include/ssal_enet.h: a handle is re-entrant -- any number of host threads may score on it, each on its own stream and
    workspace.  Two threads x two torch streams x one model object, interleaved for many calls, each preceded by work on
    its own stream that the chains must wait for (the input is produced on that stream right before the call): every
    score must equal the single-threaded result bit for bit (fork / join events are per call, ssal::ChainSet)."""
    import threading
    net, _ = enet_c3k19
    h, w = 256, 512
    firsts = [(3, 4), (200, 3)]
    want = [net.score(syn.synth_frames_device(f, n, h, w, 3), "entropy").cpu().numpy() for f, n in firsts]
    torch.cuda.synchronize()
    errors, results = [], [[], []]
    start = threading.Barrier(2)

    def worker(t):
        try:
            f, n = firsts[t]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                start.wait()
                for it in range(24):
                    x = torch.empty((n, h, w, 3), dtype=torch.float32, device="cuda")
                    x.fill_(float("nan"))                              # a chain that starts early scores NaNs
                    syn.synth_frames_device(f, n, h, w, 3, out=x)      # producer on THIS thread's stream
                    results[t].append(net.score(x, "entropy"))
                stream.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for t in range(2):
        assert len(results[t]) == 24
        for r in results[t]:
            assert np.array_equal(r.cpu().numpy(), want[t]), (t, r.cpu().numpy(), want[t])
    assert 3 <= len(net._workspaces) <= net.MAX_WORKSPACES  # main stream + one workspace per worker stream


def test_workspaces_per_stream_are_bounded(enet_c3k19):
    """one workspace per (device, stream) keeps concurrent callers apart; a caller that cycles through many streams must not
    pin one workspace per stream for ever: at most DeviceState.MAX_WORKSPACES stay alive, results unchanged"""
    net, _ = enet_c3k19
    x = syn.synth_frames_device(21, 2, 64, 128, 3)
    want = net.score(x, "entropy").cpu().numpy()
    streams = [torch.cuda.Stream() for _ in range(7)]
    for rep in range(2):
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                got = net.score(x, "entropy")
            st.synchronize()
            assert np.array_equal(got.cpu().numpy(), want)
            assert len(net._workspaces) <= net.MAX_WORKSPACES


def test_failed_call_leaves_no_chain_running(enet_c3k19):
    """a call that is refused (workspace too small) returns its status and the handle keeps working; nothing of the refused
    call is left on the side streams"""
    import ctypes
    net, _ = enet_c3k19
    x = syn.synth_frames_device(7, 2, 64, 128, 3)
    want = net.score(x, "entropy").cpu().numpy()
    L = _lib.lib()
    scores = torch.empty((2,), dtype=torch.float64, device="cuda")
    small = torch.empty((4096,), dtype=torch.uint8, device="cuda")
    rc = L.ssal_enet_score_nhwc(net._handle, _lib.dev_ptr(x), 2, 64, 128, 0, 0.0, _lib.dev_ptr(scores, torch.float64, "scores"),
                                None, None, None, _lib.dev_ptr(small), small.numel(), _lib.stream_ptr())
    assert rc == 5  # SSAL_ENOMEM
    assert np.array_equal(net.score(x, "entropy").cpu().numpy(), want)


def test_repeated_calls_do_not_grow_device_memory(enet_c3k19):
    """ENet.__call__ keeps only the most recent logits / endpoints (an eager stand-in for the reference's
    once-per-graph-build `outputs.append`, enet.py:405): memory stays flat over many calls"""
    net, _ = enet_c3k19
    x = syn.synth_frames_device(0, 2, 64, 128, 3)
    for _ in range(3):
        net(x, training=False)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(200):
        net(x, training=False)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20)
    assert len(net.outputs) == 1 and len(net.endpoint_outputs) == 1


def test_knobs_are_reported_and_default():
    k = _lib.get_knobs()
    assert k["defaults"] == 1 and k["measure_build"] == 0 and k["fuse_ends"] == 3 and k["ablate"] == 0 and "MEASUREMENT" not in k["version"]
    with pytest.raises(ValueError):
        _lib.set_knob("ablate", 1)  # no work-skipping switch in the product build
    _lib.set_knob("bnk_tw", 16)
    try:
        assert _lib.get_knobs()["defaults"] == 0
    finally:
        _lib.set_knob("bnk_tw", 0)
    assert _lib.get_knobs()["defaults"] == 1
