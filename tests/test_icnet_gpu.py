"""GPU parity tests of the ICNet row (BASELINE config C4): the HIP path through the C ABI (include/ssal_icnet.h)
against the CPU oracle (oracle/icnet_oracle.py) on identical seeded inputs.

PARITY STATUS of this row: unpinned AND undefined -- the reference's models/icnet/icnet.py:1-7 is an empty class;
ICNET_SPEC.md defines the network.  What is asserted: conv outputs, logits and labels bit-exact against the C
restatement (same accumulation order), <= 1e-4 against the independent torch restatement, confidences <= 1e-4
(north_star tolerance), per-image float64 means <= 1e-6, top-k id sets equal.
"""
import numpy as np
import pytest
import torch

import semanticsegmentationactivelearning_amd as ssal
from helpers import frames, report_diff
from oracle import enet_oracle as orc
from oracle import icnet_oracle as ico
from oracle import torch_restatement as tr
from semanticsegmentationactivelearning_amd import _lib, active_learning as al, synthetic as syn
from semanticsegmentationactivelearning_amd.models.util import conv_ops as cops

pytestmark = pytest.mark.gpu
TOL = 1e-4  # north_star: softmax / margin within 1e-4 fp32


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    _lib.lib()
    yield
    torch.cuda.synchronize()


@pytest.fixture(scope="module")
def icnet19():
    net = ssal.ICNet(19)
    net.build((None, None, None, 3))
    syn.randomize_icnet(net, seed=0)
    return net, syn.icnet_params_dict(net)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bn(rng, c):
    return (rng.normal(0, 0.1, c).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32),
            rng.uniform(0.8, 1.2, c).astype(np.float32), rng.normal(0, 0.1, c).astype(np.float32))


def _oracle_conv(x, k, stride, dil, bn, bias, res, relu, up2):
    if up2:
        x = ico.resize_bilinear(x, 2 * x.shape[1], 2 * x.shape[2])
    y = orc.conv2d_same(x, k, stride=stride, dil=dil)
    if bn is not None:
        s, t = orc.bn_fold(*bn)
        return ico.affine_add_relu(y, s, t, res, relu)
    return ico.affine_add_relu(y, None, bias, res, relu)


# ---- the fused convolution operator (every ICNet layer) ---------------------------------------------
@pytest.mark.parametrize("kh,cin,cout,stride,dil,n,h,w,res,relu,up2", [
    (1, 32, 32, 1, 1, 2, 9, 13, False, True, False),
    (1, 64, 128, 1, 1, 1, 16, 20, True, True, False),
    (1, 128, 64, 2, 1, 2, 10, 14, False, False, False),     # stride-2 1x1 (conv3_1 reduce / proj)
    (3, 32, 32, 1, 1, 1, 12, 17, False, True, False),
    (3, 32, 64, 1, 1, 2, 7, 9, False, True, False),
    (3, 32, 32, 2, 1, 1, 16, 24, False, True, False),       # conv2_sub1
    (3, 32, 32, 2, 1, 1, 15, 11, False, True, False),       # odd dims: SAME pads (1, 1)
    (3, 32, 64, 2, 1, 2, 36, 50, False, True, False),       # conv3_sub1's shape: two column tiles, ragged 18 x 25 output
    (3, 32, 19, 2, 1, 1, 2, 2, False, False, False),        # one output pixel, 19 of 32 columns, no ReLU
    (3, 64, 64, 1, 1, 1, 6, 8, False, True, False),
    (3, 128, 128, 1, 2, 1, 12, 16, False, True, False),     # dilation 2 (conv4)
    (3, 256, 256, 1, 4, 1, 9, 10, False, True, False),      # dilation 4 (conv5)
    (1, 1024, 256, 1, 1, 1, 4, 8, False, True, False),
    (1, 256, 1024, 1, 1, 1, 4, 6, True, True, False),
    (3, 256, 128, 1, 2, 1, 5, 7, True, True, True),         # conv_sub4: 2x interp inside the conv + fusion add
    (3, 128, 128, 1, 2, 2, 6, 9, True, True, True),         # conv_sub2
    (3, 128, 128, 1, 2, 2, 6, 10, True, True, True),        # Wo % 4 == 0: the four-adjacent-pixels loader (k_igemm<NT, 2>), all four borders
    (3, 256, 128, 1, 2, 1, 5, 8, False, True, True),
    (3, 32, 64, 1, 4, 1, 7, 6, False, False, True),         # dilation 4, padding 4, negative inputs kept (no ReLU upstream)
    (3, 64, 96, 1, 2, 3, 1, 2, True, False, True),          # one source row, two source columns: every clamp at once
    (1, 128, 19, 1, 1, 1, 7, 5, False, False, True),        # conv6_cls: 2x interp + bias, 19 of 32 columns
    (1, 32, 5, 1, 1, 3, 3, 3, False, False, False),
    (3, 32, 32, 1, 1, 1, 130, 3, False, True, False),       # more pixels than one 128-row tile, ragged tail
])
def test_conv_bn_act_bit_exact(kh, cin, cout, stride, dil, n, h, w, res, relu, up2):
    rng = np.random.default_rng(kh * 1000 + cin + cout + h)
    x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
    k = (rng.normal(size=(kh, kh, cin, cout)) / np.sqrt(kh * kh * cin)).astype(np.float32)
    use_bias = cout in (19, 5)
    bn = None if use_bias else _bn(rng, cout)
    bias = rng.normal(size=cout).astype(np.float32) if use_bias else None
    hh, ww = (2 * h, 2 * w) if up2 else (h, w)
    oh, ow = -(-hh // stride), -(-ww // stride)
    r = rng.normal(size=(n, oh, ow, cout)).astype(np.float32) if res else None
    want = _oracle_conv(x, k, stride, dil, bn, bias, r, relu, up2)
    got = cops.conv_bn_act(dev(x), k, stride, dil, bn=bn, bias=bias, residual=None if r is None else dev(r),
                           relu=relu, upsample2x=up2)
    report_diff("conv_bn_act", got.cpu().numpy(), want)


@pytest.mark.parametrize("cin,h,w", [(3, 16, 24), (3, 17, 9), (1, 8, 8), (4, 10, 12)])
def test_first_conv_bit_exact(cin, h, w):
    rng = np.random.default_rng(cin + h)
    x = rng.uniform(0, 1, size=(2, h, w, cin)).astype(np.float32)
    k = (rng.normal(size=(3, 3, cin, 32)) / 5).astype(np.float32)
    bn = _bn(rng, 32)
    want = _oracle_conv(x, k, 2, 1, bn, None, None, True, False)
    got = cops.conv_bn_act(dev(x), k, 2, 1, bn=bn, relu=True)
    report_diff("first conv", got.cpu().numpy(), want)


def test_conv_argument_errors():
    x = torch.zeros((1, 4, 4, 24), device="cuda")
    with pytest.raises(ValueError):
        cops.conv_bn_act(x, np.zeros((3, 3, 24, 32), np.float32))  # cin % 32 != 0 and not a first layer
    with pytest.raises(ValueError):
        cops.conv_bn_act(torch.zeros((1, 4, 4, 32), device="cuda"), np.zeros((3, 3, 64, 32), np.float32))


@pytest.mark.parametrize("n,h,w,c", [(2, 16, 24, 64), (1, 9, 7, 8), (1, 2, 2, 4)])
def test_max_pool_3x3_s2(n, h, w, c):
    x = np.random.default_rng(h).normal(size=(n, h, w, c)).astype(np.float32)
    report_diff("maxpool", cops.max_pool_3x3_s2(dev(x)).cpu().numpy(), ico.maxpool3x3_s2(x))


@pytest.mark.parametrize("n,h,w,c", [(2, 32, 64, 64), (1, 7, 9, 8), (1, 2, 4, 1024), (1, 1, 1, 4)])
def test_pyramid_pooling(n, h, w, c):
    x = np.random.default_rng(w).normal(size=(n, h, w, c)).astype(np.float32)
    report_diff("ppm", cops.pyramid_pooling(dev(x)).cpu().numpy(), ico.pyramid_pooling(x))


@pytest.mark.parametrize("measure", ["margin", "entropy", "confidence"])
@pytest.mark.parametrize("n,h,w,k", [(2, 8, 12, 19), (1, 5, 3, 6), (1, 16, 16, 2)])
def test_upscore_kernel(measure, n, h, w, k):
    """4x bilinear + score fused == resize_bilinear then score, bit-exact labels, conf within 1e-4"""
    lq = (np.random.default_rng(k + h).normal(size=(n, h, w, k)) * 4).astype(np.float32)
    full = ico.resize_bilinear(lq, 4 * h, 4 * w)
    want_mean, want_conf, want_label = orc.score_logits(full, measure)
    thr = float(np.median(want_conf))
    s, e = cops.upscore_logits(dev(lq), measure, threshold=thr, return_label=True, return_mask=True,
                               return_confidence=True)
    report_diff("label", e["label"].cpu().numpy(), want_label)
    conf = e["confidence"].cpu().numpy()
    report_diff("confidence", conf, want_conf, exact=False, atol=TOL)
    report_diff("mean", s.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    assert (e["mask"].cpu().numpy() == (conf >= np.float32(thr))).all()
    # score-only launch (no per-pixel outputs) gives the same means
    assert torch.equal(s, cops.upscore_logits(dev(lq), measure))


# ---- whole network --------------------------------------------------------------------------------
def _check_net(net, P, x, tag, every_endpoint=True):
    ep = {}
    want = ico.icnet_forward(P, x, ep)
    got = net(dev(x), training=False).cpu().numpy()
    if every_endpoint:
        names = net.endpoint_names()
        assert len(names) > 60
        for nm in names:
            report_diff("%s %s" % (tag, nm), net.endpoint(nm).cpu().numpy(), ep[nm])
    report_diff(tag + " logits vs C oracle (bit-exact)", got, want)
    return got, want


def test_every_block_and_logits_bit_exact(icnet19):
    """every materialised ICNET_SPEC layer output (67 tensors) + the logits, 2 frames of 64x128"""
    net, P = icnet19
    x = frames([0, 1], 64, 128, 3)
    got, want = _check_net(net, P, x, "64x128")
    wb = tr.icnet_forward(P, x)
    report_diff("logits vs torch restatement", got, wb, exact=False, atol=TOL)


@pytest.mark.parametrize("n,h,w", [(1, 32, 32), (3, 96, 64), (1, 160, 224)])
def test_forward_ragged_shapes(icnet19, n, h, w):
    net, P = icnet19
    _check_net(net, P, frames(list(range(20, 20 + n)), h, w, 3), "%dx%dx%d" % (n, h, w), every_endpoint=(h < 100))


@pytest.mark.parametrize("measure", ["margin", "entropy", "confidence"])
def test_score_matches_oracle(icnet19, measure):
    net, P = icnet19
    x = frames([3, 4, 5], 64, 96, 3)
    want_mean, want_conf, want_label, _ = ico.score_images(P, x, measure)
    s, e = net.score(dev(x), measure, return_label=True, return_confidence=True)
    report_diff("label (bit-exact argmax)", e["label"].cpu().numpy(), want_label)
    report_diff("confidence", e["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL)
    report_diff("mean", s.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    # fused score == score of the materialised logits (same per-pixel bits; the float64 block sums differ in order)
    s2, e2 = al.score_logits(net(dev(x), training=False), measure, return_confidence=True)
    assert torch.equal(e["confidence"], e2["confidence"])
    assert torch.allclose(s, s2, rtol=0, atol=1e-12)
    # bitwise reproducible
    assert torch.equal(s, net.score(dev(x), measure))


def test_uint8_frames_and_other_class_counts():
    net = ssal.ICNet(6)
    net.build((None, None, None, 4))
    syn.randomize_icnet(net, seed=3)
    P = syn.icnet_params_dict(net)
    xu = syn.synth_frames_device(7, 2, 64, 64, 4, dtype=torch.uint8)
    xf = syn.synth_frames_device(7, 2, 64, 64, 4)
    a, b = net(xu, training=False), net(xf, training=False)
    assert torch.equal(a, b)
    report_diff("4-channel / 6-class logits", b.cpu().numpy(), ico.icnet_forward(P, xf.cpu().numpy()))
    assert torch.equal(net.score(xu, "margin"), net.score(xf, "margin"))


def test_rank_confidence_margin_topk_ids_match_oracle(icnet19):
    """config C4 end to end at a size the oracle finishes in seconds: ICNet + margin + top-k over a pool"""
    net, P = icnet19
    ids = list(range(40, 52))
    h, w, bs, k = 64, 64, 4, 5
    batches = [(syn.synth_frames_device(ids[i], bs, h, w, 3), np.arange(i, i + bs)) for i in range(0, len(ids), bs)]
    unlabelled = np.array([0, 1, 2, 4, 5, 7, 8, 9, 10, 11])
    low, uconf = al.rank_confidence(net, batches, len(ids), unlabelled, k, measure="margin")
    want_mean = ico.score_images(P, frames(ids, h, w, 3), "margin")[0]
    want_low, want_u = orc.rank_lowest(want_mean, unlabelled, k)
    srt = np.sort(want_u)
    assert srt[k] - srt[k - 1] > 1e-4, "fixture must separate the k-th boundary"
    assert set(low.tolist()) == set(want_low.tolist())
    report_diff("unlabelled confidence", uconf, want_u, exact=False, atol=1e-6)


def test_weight_update_and_errors(icnet19):
    net = ssal.ICNet(19)
    net.build((None, None, None, 3))
    syn.randomize_icnet(net, seed=9)
    x = syn.synth_frames_device(0, 1, 32, 64, 3)
    a = net(x, training=False).clone()
    net.conv6_cls.bias.assign(net.conv6_cls.bias.numpy() + 1.0)
    b = net(x, training=False)
    assert torch.allclose(a + 1.0, b, atol=1e-5)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 40, 64, 3), device="cuda"), training=False)
    with pytest.raises(NotImplementedError):
        net(x, training=True)
    with pytest.raises(NotImplementedError):
        net.score(x, "bald")
    with pytest.raises(ValueError):
        net.endpoint("sub24_sum_interp")  # evaluated inside conv_sub2, never materialised


def test_full_resolution_frame_bit_exact(icnet19):
    """one 1024x2048 frame (config C4's size): 1/4-resolution class scores bit-exact, labels bit-exact, margin
    mean within 1e-6 of the oracle"""
    net, P = icnet19
    x = frames([2], 1024, 2048, 3)
    ep = {}
    want_q = ico.icnet_forward(P, x, ep, full_logits=False)
    s, e = net.score(dev(x), "margin", return_label=True)
    report_diff("conv6_cls @1024x2048", net.endpoint("conv6_cls").cpu().numpy(), want_q)
    report_diff("sub24_sum @1024x2048", net.endpoint("sub24_sum").cpu().numpy(), ep["sub24_sum"])
    full = ico.resize_bilinear(want_q, 1024, 2048)
    want_mean, _, want_label = orc.score_logits(full, "margin")
    report_diff("label", e["label"].cpu().numpy(), want_label)
    report_diff("margin mean", s.cpu().numpy(), want_mean, exact=False, atol=1e-6)
    from helpers import pool_score_table
    table = pool_score_table("icnet", 3, 19, 1024, 2048, "margin", 0)
    assert s.cpu().numpy()[0] == table[2], "HIP score of ICNet frame 2 != committed pool_scores.npz entry"


def test_matches_golden_fixture(icnet19):
    import os
    net, P = icnet19
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "icnet_c3k19_64x128.npz"))
    x = frames(list(g["frame_ids"]), 64, 128, 3)
    for m in ("margin", "entropy", "confidence"):
        s, e = net.score(dev(x), m, return_label=True, return_confidence=True)
        report_diff(m + " label vs golden", e["label"].cpu().numpy(), g["label"])
        report_diff(m + " conf vs golden", e["confidence"].cpu().numpy()[0], g["conf_" + m], exact=False, atol=TOL)
        report_diff(m + " mean vs golden", s.cpu().numpy(), g["mean_" + m], exact=False, atol=1e-6)
    report_diff("1/4-resolution logits vs golden", net.endpoint("conv6_cls").cpu().numpy(), g["logits_quarter"])
    report_diff("sub12_sum slice vs golden", net.endpoint("sub12_sum").cpu().numpy()[0, :4, :4, :], g["sub12_sum_slice"])


def test_icnet_from_tfrecords_and_inference_path(icnet19, tmp_path):
    """the callers on either side of the path take ICNet unchanged: TFRecords -> InputStage (uint8 frames, pinned
    batches, side-stream copy) -> ICNet margin ranking; and inference.predict_labels / run_inference"""
    import os
    from PIL import Image
    from test_input_cpu import write_pool
    from semanticsegmentationactivelearning_amd import inference as inf
    from semanticsegmentationactivelearning_amd.tensortools import InputStage, NumpyCapsule
    net, P = icnet19
    num, k = 6, 2
    files = write_pool(str(tmp_path), num, 64, 64, with_label=False)
    cap = NumpyCapsule(shuffle=True, seed=5)
    cap.filenames, cap.indices = files, np.arange(num)
    cap.labelled = np.zeros(num, dtype=bool)
    stage = InputStage(input_shape=[64, 64], seed=6, image_dtype=np.uint8, pin_memory=True)
    stage.add_dataset_from_placeholders("train", cap.filenames, cap.labelled, cap.indices, batch_size=4)
    stage.init_iterator("train", None, cap.feed_dict)

    def batches():
        for image, label, mask, labelled, index in stage:
            yield image, index

    unlabelled = np.arange(num)
    low, uc = al.rank_confidence(net, batches(), num, unlabelled, k, measure="margin", prefetch=2)
    want = np.concatenate([ico.score_images(P, frames([i], 64, 64, 3), "margin")[0] for i in range(num)])
    want_low, want_uc = orc.rank_lowest(want, unlabelled, k)
    assert set(low.tolist()) == set(want_low.tolist())
    report_diff("unlabelled_confidence", uc, want_uc, exact=False, atol=1e-6)
    # inference.py path (reference inference.py:95-119)
    x = frames([30, 31], 64, 96, 3)
    want_logits = ico.icnet_forward(P, x)
    pred = inf.predict_labels(net, dev(x))
    report_diff("trainId map", pred.cpu().numpy(), want_logits.argmax(-1).astype(np.uint8))
    paths = inf.run_inference(net, [(x, [b"a", "b"])], str(tmp_path / "out"))
    assert [os.path.basename(p) for p in paths] == ["a.png", "b.png"]
    assert (np.asarray(Image.open(paths[0])) == want_logits[0].argmax(-1)).all()


def test_conv_bn_act_random_shapes_bit_exact():
    """seeded sweep over the fused convolution's argument space (kernel size, channel counts incl. non-multiples of 32
    on the output side, stride, dilation, ragged spatial sizes, shortcut add, ReLU, on-the-fly 2x interpolation): every
    case bit-exact against the C oracle.  Covers the dispatch between k_igemm<1|2|4>, k_conv3x3_c32 and
    k_conv1x1_up2_c128."""
    rng = np.random.default_rng(20260)
    cases = 0
    for _ in range(48):
        kh = int(rng.choice([1, 3]))
        cin = int(rng.choice([32, 32, 64, 96, 128, 160]))
        cout = int(rng.choice([1, 7, 19, 32, 33, 64, 96, 100, 128, 130, 256]))
        stride = int(rng.choice([1, 1, 2]))
        dil = int(rng.choice([1, 1, 2, 3, 4])) if kh == 3 else 1
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 29)), int(rng.integers(1, 37))
        up2 = bool(rng.integers(0, 4) == 0) and stride == 1
        res = bool(rng.integers(0, 2))
        relu = bool(rng.integers(0, 2))
        x = rng.normal(size=(n, h, w, cin)).astype(np.float32)
        k = (rng.normal(size=(kh, kh, cin, cout)) / np.sqrt(kh * kh * cin)).astype(np.float32)
        bn = _bn(rng, cout)
        hh, ww = (2 * h, 2 * w) if up2 else (h, w)
        oh, ow = -(-hh // stride), -(-ww // stride)
        r = rng.normal(size=(n, oh, ow, cout)).astype(np.float32) if res else None
        want = _oracle_conv(x, k, stride, dil, bn, None, r, relu, up2)
        got = cops.conv_bn_act(dev(x), k, stride, dil, bn=bn, residual=None if r is None else dev(r), relu=relu,
                               upsample2x=up2)
        report_diff("case k%d cin%d cout%d s%d d%d n%d %dx%d up2=%s res=%s" % (kh, cin, cout, stride, dil, n, h, w, up2, res),
                    got.cpu().numpy(), want)
        cases += 1
    assert cases == 48


@pytest.mark.parametrize("seed", [1, 2])
def test_icnet_random_sizes_and_batches(icnet19, seed):
    """whole network at random multiples of 32 (non-square, tiny and mid sizes) and batch sizes: logits bit-exact"""
    net, P = icnet19
    rng = np.random.default_rng(seed)
    for _ in range(3):
        n = int(rng.integers(1, 4))
        h, w = 32 * int(rng.integers(1, 6)), 32 * int(rng.integers(1, 8))
        x = frames(list(range(60, 60 + n)), h, w, 3)
        want = ico.icnet_forward(P, x)
        got = net(dev(x), training=False).cpu().numpy()
        report_diff("ICNet %dx%dx%d" % (n, h, w), got, want)
        s = net.score(dev(x), "margin")
        wm = orc.score_logits(want, "margin")[0]
        report_diff("margin mean %dx%dx%d" % (n, h, w), s.cpu().numpy(), wm, exact=False, atol=1e-6)


def test_repeated_scoring_is_bitwise_stable_at_bench_tile_counts(icnet19):
    """soak: 25 back-to-back scoring passes of the same 2 x 512 x 1024 batch (every kernel runs with thousands of
    workgroups, the implicit-GEMM loops with up to 72 chunks and LDS double buffers written from inside the matrix
    section) must give the same float64 scores and the same per-pixel labels bit for bit -- a race between a wave still
    reading an LDS buffer and another one refilling it would show up here as a flipped pixel"""
    net, _ = icnet19
    x = syn.synth_frames_device(300, 2, 512, 1024, 3)
    s0, e0 = net.score(x, "margin", return_label=True, return_confidence=True)
    s0, l0, c0 = s0.clone(), e0["label"].clone(), e0["confidence"].clone()
    for _ in range(24):
        s, e = net.score(x, "margin", return_label=True, return_confidence=True)
        assert torch.equal(s, s0) and torch.equal(e["label"], l0) and torch.equal(e["confidence"], c0)


def test_image_group_streams_are_bit_identical(icnet19):
    """ICNet's score path may run as `ic_groups` image chains on library-owned side streams (default 1 since round 5: one chain
    is faster with the three-workgroup up-sampling kernel; ENet's knob is `img_groups`): scores and labels must not depend on the
    grouping (1, 2, 3 with uneven groups), also for a batch of one"""
    from semanticsegmentationactivelearning_amd import _lib
    net, _ = icnet19
    x = syn.synth_frames_device(7, 4, 64, 128, 3)
    ref = None
    try:
        assert _lib.get_knobs()["ic_groups"] == 1
        for g in (1, 2, 3):
            _lib.set_knob("ic_groups", g)
            outs = []
            for _ in range(2):
                s, e = net.score(x, "margin", return_label=True)
                outs.append((s.cpu().numpy(), e["label"].cpu().numpy(), net.score(x[:1], "margin").cpu().numpy()))
            assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[1]))
            ref = ref or outs[0]
            assert all(np.array_equal(a, b) for a, b in zip(outs[0], ref)), "ic_groups=%d changes the result" % g
    finally:
        _lib.set_knob("ic_groups", 1)


def test_two_host_threads_share_one_handle_and_family_switch(icnet19):
    """include/ssal_icnet.h: a handle is re-entrant per (stream, workspace).  Two host threads x two torch streams on ONE
    ICNet model, inputs produced on the thread's own stream right before each call; then ssal_set_kernel_family(0) -- the
    per-layer launches on the caller's stream only -- gives the same bits (the switch reaches ICNet's fused blocks)"""
    import threading
    net, _ = icnet19
    h, w = 128, 256
    firsts = [(11, 4), (300, 3)]
    want = [net.score(syn.synth_frames_device(f, n, h, w, 3), "margin").cpu().numpy() for f, n in firsts]
    torch.cuda.synchronize()
    errors, results = [], [[], []]
    start = threading.Barrier(2)

    def worker(t):
        try:
            f, n = firsts[t]
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                start.wait()
                for it in range(12):
                    x = torch.empty((n, h, w, 3), dtype=torch.float32, device="cuda")
                    x.fill_(float("nan"))
                    syn.synth_frames_device(f, n, h, w, 3, out=x)
                    results[t].append(net.score(x, "margin"))
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
        for r in results[t]:
            assert np.array_equal(r.cpu().numpy(), want[t]), (t, r.cpu().numpy(), want[t])
    try:
        _lib.lib().ssal_set_kernel_family(0)
        for (f, n), wnt in zip(firsts, want):
            assert np.array_equal(net.score(syn.synth_frames_device(f, n, h, w, 3), "margin").cpu().numpy(), wnt)
    finally:
        _lib.lib().ssal_set_kernel_family(1)


def test_interleaved_models_batches_and_shapes_match_the_single_stream_unfused_path(icnet19):
    """soak: ENet (3- and 4-channel handles) and ICNet share ONE process-wide pool of side streams; 40 back-to-back calls
    that interleave the models, the entry points (score / score with labels / forward), batch sizes 1..8 and frame sizes
    must give exactly what the same sequence gives with everything on the caller's stream and every fusion off"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_model
    from semanticsegmentationactivelearning_amd import _lib
    icn, _ = icnet19
    e3, _ = make_model(19, 3, seed=0)
    e4, _ = make_model(6, 4, seed=1)
    rng = np.random.default_rng(5)
    plan = []
    for _ in range(40):
        kind = rng.choice(["e3", "e4", "icn", "e3_label", "e3_forward"])
        n = int(rng.integers(1, 9))
        h, w = [(64, 128), (32, 96), (96, 64), (128, 160)][int(rng.integers(0, 4))]
        plan.append((kind, n, h, w, int(rng.integers(0, 2000)), ["entropy", "margin", "confidence"][int(rng.integers(0, 3))]))

    def run():
        out = []
        for kind, n, h, w, first, measure in plan:
            if kind == "icn":
                out.append(icn.score(syn.synth_frames_device(first, n, h, w, 3), measure))
            elif kind == "e4":
                out.append(e4.score(syn.synth_frames_device(first, n, h, w, 4), measure))
            elif kind == "e3_label":
                s, e = e3.score(syn.synth_frames_device(first, n, h, w, 3), measure, return_label=True)
                out += [s, e["label"]]
            elif kind == "e3_forward":
                out.append(e3(syn.synth_frames_device(first, n, h, w, 3), training=False)[:, ::7, ::5, :].clone())
            else:
                out.append(e3.score(syn.synth_frames_device(first, n, h, w, 3), measure))
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in out]

    got = run()
    again = run()
    try:
        _lib.set_knob("img_groups", 1)
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("fuse_ends", 0)
        want = run()
        _lib.set_knob("img_groups", 2)
        _lib.set_knob("ic_groups", 2)  # ICNet on two chains next to ENet's two (not the shipped default any more): same bits
        _lib.set_knob("fuse_ends", 3)
        two = run()
        assert all(np.array_equal(a, b) for a, b in zip(two, want))
    finally:
        _lib.set_knob("img_groups", 2)
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("fuse_ends", 3)
    assert len(got) == len(want)
    for i, (a, b, c) in enumerate(zip(got, want, again)):
        assert np.array_equal(a, b), "call %d of the plan differs from the single-stream, unfused result" % i
        assert np.array_equal(a, c), "call %d of the plan is not reproducible" % i


@pytest.mark.parametrize("n,h,w,u8", [(3, 64, 96, False), (1, 32, 32, False), (2, 160, 224, True), (5, 96, 64, False)])
def test_fused_branch_fronts_are_bit_identical(icnet19, n, h, w, u8):
    """knob ic_front: conv1_sub1 + conv2_sub1 (bit 0) and conv1_1_3x3_s2 + conv1_2_3x3 (bit 1) as ONE launch each
    (k_front2, csrc/ssal_icnet_front.hip; the first convolution's output never reaches HBM).  The second convolution's
    output must equal the per-layer path's bit for bit (ragged tiles, image borders, uint8 frames), and so must scores
    and labels, with one and with two image chains."""
    net, _ = icnet19
    x = syn.synth_frames_device(11, n, h, w, 3)
    if u8:
        x = (x * 255.0).round().clamp(0, 255).to(torch.uint8)
    shipped = _lib.get_knobs()["ic_front"]
    try:
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("ic_front", 0)
        s0, e0 = net.score(x, "margin", return_label=True, return_confidence=True)
        want = {k: net.endpoint(k).clone() for k in ("conv2_sub1", "conv1_2_3x3")}
        for front in (1, 2, 3):
            _lib.set_knob("ic_front", front)
            for groups in (1, 2):
                _lib.set_knob("ic_groups", groups)
                s, e = net.score(x, "margin", return_label=True, return_confidence=True)
                assert torch.equal(s, s0) and torch.equal(e["label"], e0["label"]) and torch.equal(e["confidence"], e0["confidence"]), \
                    "ic_front=%d ic_groups=%d changes the result" % (front, groups)
                if groups == 1:
                    for k, t in want.items():
                        assert torch.equal(net.endpoint(k), t), "ic_front=%d: %s differs" % (front, k)
    finally:
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("ic_front", shipped)


@pytest.mark.parametrize("n,h,w", [(2, 64, 96), (1, 32, 32), (3, 160, 224)])
def test_projection_shortcut_inside_the_increase_launch_is_bit_identical(icnet19, n, h, w):
    """knob ic_dual: the four blocks with a projection shortcut (conv2_1, conv3_1 -- stride 2 --, conv4_1, conv5_1) evaluate
    the projection inside their 1x1 increase launch (k_igemm<.., DUAL>) instead of writing it and reading it back: block
    outputs, scores, labels and confidences must be the same bits, with one and with two image chains."""
    net, _ = icnet19
    x = syn.synth_frames_device(21, n, h, w, 3)
    shipped = _lib.get_knobs()["ic_dual"]
    blocks = ("conv2_1", "conv3_1", "conv4_1", "conv5_1", "conv5_3")
    try:
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("ic_dual", 0)
        s0, e0 = net.score(x, "margin", return_label=True, return_confidence=True)
        want = {k: net.endpoint(k).clone() for k in blocks}
        _lib.set_knob("ic_dual", 1)
        for groups in (1, 2):
            _lib.set_knob("ic_groups", groups)
            s, e = net.score(x, "margin", return_label=True, return_confidence=True)
            assert torch.equal(s, s0) and torch.equal(e["label"], e0["label"]) and torch.equal(e["confidence"], e0["confidence"]), \
                "ic_dual=1 ic_groups=%d changes the result" % groups
            if groups == 1:
                for k, t in want.items():
                    assert torch.equal(net.endpoint(k), t), "ic_dual=1: %s differs" % k
    finally:
        _lib.set_knob("ic_groups", 1)
        _lib.set_knob("ic_dual", shipped)


@pytest.mark.parametrize("n,h,w", [(2, 64, 96), (1, 160, 224)])
def test_single_lds_buffer_igemm_is_bit_identical(icnet19, n, h, w):
    """knob ig_sb: k_igemm with ONE LDS buffer at three workgroups per CU (3, shipped: the up-sampling form and the plain form
    at NT = 1 / 2; 2: without NT = 1; 1: the up-sampling form only; 0: double-buffered everywhere): block outputs, logits and
    scores are the same bits"""
    net, _ = icnet19
    x = syn.synth_frames_device(33, n, h, w, 3)
    shipped = _lib.get_knobs()["ig_sb"]
    assert shipped == 3
    try:
        _lib.set_knob("ig_sb", 0)
        want_logits = net(x, training=False).clone()
        want = {k: net.endpoint(k).clone() for k in ("sub24_sum", "sub12_sum", "conv5_4_k1", "conv3_1")}
        s0 = net.score(x, "margin")
        for sb in (1, 2, 3):
            _lib.set_knob("ig_sb", sb)
            assert torch.equal(net(x, training=False), want_logits), "ig_sb=%d changes the logits" % sb
            for k, t in want.items():
                assert torch.equal(net.endpoint(k), t), "ig_sb=%d: %s differs" % (sb, k)
            assert torch.equal(net.score(x, "margin"), s0)
    finally:
        _lib.set_knob("ig_sb", shipped)


def test_endpoint_after_score_raises_for_layers_inside_fused_launches(icnet19):
    """score() runs fused launches that never write some ICNET_SPEC layer outputs; endpoint() of such a name
    after a score() must raise instead of returning a stale slice of the torch.empty workspace.  After a forward every
    endpoint is valid again; the C answer (ssal_icnet_endpoint_valid_after_score) follows the knobs."""
    net, _ = icnet19
    x = dev(frames([0], 64, 64, 3))
    net(x, training=False)
    assert net.endpoint("conv1_sub1").shape == (1, 32, 32, 32)  # forward: materialised
    net.score(x, measure="margin")
    swallowed = [nm for nm in net.endpoint_names()
                 if _lib.lib().ssal_icnet_endpoint_valid_after_score(net._handle, nm.encode(), 64, 64) == 0]
    assert "conv1_sub1" in swallowed and "conv2_1_1x1_proj" in swallowed and "conv2_2_3x3" in swallowed
    assert "conv1_1_3x3_s2" in swallowed  # both branch fronts run as one launch each since round 5 (ic_front = 3)
    assert "conv6_cls" not in swallowed and "conv2_2" not in swallowed and "conv2_sub1" not in swallowed
    for nm in swallowed:
        with pytest.raises(RuntimeError, match="not written by score"):
            net.endpoint(nm)
    assert net.endpoint("conv6_cls").shape == (1, 16, 16, 19)
    shipped_front = _lib.get_knobs()["ic_front"]
    try:  # with the fused front and the dual launches off, those buffers ARE written by a score call
        _lib.set_knob("ic_front", 0)
        _lib.set_knob("ic_dual", 0)
        net.score(x, measure="margin")
        assert net.endpoint("conv1_sub1").shape == (1, 32, 32, 32) and net.endpoint("conv2_1_1x1_proj").shape[-1] == 128
    finally:
        _lib.set_knob("ic_front", shipped_front)
        _lib.set_knob("ic_dual", 1)
    assert _lib.lib().ssal_icnet_endpoint_valid_after_score(net._handle, b"no_such_layer", 64, 64) == -1
    net(x, training=False)
    for nm in swallowed:
        net.endpoint(nm)
