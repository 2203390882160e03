"""Parity gate of the OPT-IN arithmetic mode ``arithmetic="bf16x3"`` (include/ssal_enet.h SSAL_ARITH_BF16X3;
csrc/ssal_bottleneck_bf16x3.hip): the sixteen 128-channel regular / dilated / asymmetric bottlenecks, the downsample block Bottleneck2_0 and the upsample block Bottleneck4_0 evaluate their
convolutions on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into three bf16 terms (six cross products, fp32
accumulation).  It is a different summation than the oracle's fmaf chains, so it is NOT bit-identical to the default
mode and has its own gate -- north_star's tolerance: per-pixel softmax / entropy / margin within 1e-4, identical top-k
example ids; plus, asserted here: pooling indices bit-identical (the first pooling layer runs in front of these blocks; Bottleneck2_0's own max-pool +
argmax is evaluated in exact fp32 inside its split-operand kernel), per-image float64 scores within 1e-6, logits within 1e-4 of the C oracle.

The default mode ("f32") stays the reference's arithmetic, the bench headline and what every other test checks.
PARITY STATUS as everywhere: the oracle is this repository's restatement of the reference (TensorFlow is not
installable here, the reference ships no fixtures): parity with TensorFlow itself is unpinned."""
import numpy as np
import pytest
import torch

from helpers import frames, pool_score_table, report_diff
from oracle import enet_oracle as orc
from semanticsegmentationactivelearning_amd import _lib, active_learning as al, synthetic as syn

pytestmark = pytest.mark.gpu
TOL_CONF = 1e-4   # north_star: softmax / entropy within 1e-4 fp32
TOL_SCORE = 1e-6  # per-image float64 mean
TOL_LAYER = 2e-5  # one block's output (|y| <= ~7) against the oracle's exact-fp32 chain


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    _lib.lib()
    yield
    torch.cuda.synchronize()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name,n,h,w", [
    ("Bottleneck2_1", 2, 16, 32), ("Bottleneck2_1", 1, 9, 11), ("Bottleneck2_1", 1, 8, 40), ("Bottleneck2_1", 1, 1, 1),
    ("Bottleneck2_2", 1, 17, 35), ("Bottleneck2_4", 2, 12, 9), ("Bottleneck2_6", 1, 18, 20), ("Bottleneck2_8", 1, 34, 36),
    ("Bottleneck3_8", 2, 32, 64), ("Bottleneck3_5", 3, 24, 72), ("Bottleneck2_3", 2, 16, 32), ("Bottleneck2_7", 1, 9, 11),
    ("Bottleneck3_3", 1, 24, 72), ("Bottleneck3_7", 1, 8, 40), ("Bottleneck2_7", 1, 5, 3), ("Bottleneck2_3", 1, 1, 1),
    ("Bottleneck3_1", 2, 128, 256), ("Bottleneck3_7", 1, 128, 256), ("Bottleneck3_8", 1, 128, 256),
])
def test_split_operand_blocks_match_the_oracle_within_tolerance(enet_c3k19, name, n, h, w):
    """every kind of block the mode has a kernel for (3x3 at dilation 1 / 2 / 4 / 8 / 16, (5,1)+(1,5)), ragged shapes, border
    tiles, partial phase sub-images, and the bench shape: within TOL_LAYER of the oracle's exact-fp32 block"""
    net, P = enet_c3k19
    layer = getattr(net, name)
    x = np.random.default_rng(18).normal(size=(n, h, w, 128)).astype(np.float32)
    want = orc.bottleneck(P, name, x, dil=layer.dilation_rate[0], asym=layer.asymmetric)
    got = layer(dev(x), training=False, arithmetic="bf16x3").cpu().numpy()
    report_diff(name + " bf16x3 vs oracle", got, want, exact=False, atol=TOL_LAYER)
    exact = layer(dev(x), training=False).cpu().numpy()
    report_diff(name + " default mode still bit-exact", exact, want)
    assert not np.array_equal(got, exact) or got.size < 256, "the opt-in mode produced the exact kernel's bits: not dispatched?"


@pytest.mark.parametrize("n,h,w", [(2, 32, 64), (1, 18, 22), (1, 16, 80), (1, 2, 2), (3, 34, 70), (1, 256, 512)])
def test_split_operand_downsample_block(enet_c3k19, n, h, w):
    """Bottleneck2_0 (64 -> 128, 2x2 / s2 projection) under the mode: output within TOL_LAYER of the oracle's exact-fp32 block;
    the max-pool residual and its window codes are exact fp32 whatever the mode -- the int64 argmax tensor is the oracle's bit
    for bit; ragged sizes (tiles that cross the border, a single output pixel) and the bench shape"""
    net, P = enet_c3k19
    x = np.random.default_rng(h + w).normal(size=(n, h, w, 64)).astype(np.float32)
    want, want_arg = orc.bottleneck_down(P, "Bottleneck2_0", x)
    got, arg = net.Bottleneck2_0(dev(x), training=False, arithmetic="bf16x3")
    report_diff("Bottleneck2_0 bf16x3 argmax (bit-identical)", arg.cpu().numpy(), want_arg)
    report_diff("Bottleneck2_0 bf16x3 vs oracle", got.cpu().numpy(), want, exact=False, atol=TOL_LAYER)
    exact, arg0 = net.Bottleneck2_0(dev(x), training=False)
    report_diff("Bottleneck2_0 default mode still bit-exact", exact.cpu().numpy(), want)
    assert torch.equal(arg0, arg)
    assert not torch.equal(got, exact) or got.numel() < 1024, "the opt-in mode produced the exact kernel's bits: not dispatched?"


@pytest.mark.parametrize("n,h,w", [(2, 16, 32), (1, 9, 11), (1, 8, 40), (1, 1, 1), (3, 17, 35), (1, 128, 256)])
def test_split_operand_upsample_block(enet_c3k19, n, h, w):
    """Bottleneck4_0 (128 -> 64; projection, transposed 3x3 / s2 convolution per output parity, expansion, 1x1 residual
    convolution + unpool_2d by the pooling indices) under the mode: within TOL_LAYER of the oracle's exact-fp32 block, with real
    pooling indices (every value in its own 2x2 window, as Bottleneck1_0 / 2_0 produce them); ragged sizes and the bench shape"""
    net, P = enet_c3k19
    rng = np.random.default_rng(h * 7 + w)
    x = rng.normal(size=(n, h, w, 128)).astype(np.float32)
    # indices of a 2x2 / s2 max-pool of a random [n, 2h, 2w, 64] tensor: flat (y * W + x) * C + c per image, as the reference's
    big = rng.normal(size=(n, 2 * h, 2 * w, 64)).astype(np.float32)
    win = big.reshape(n, h, 2, w, 2, 64).transpose(0, 1, 3, 5, 2, 4).reshape(n, h, w, 64, 4)
    code = win.argmax(-1)  # first maximum in (dy, dx) order
    yy = 2 * np.arange(h)[None, :, None, None] + code // 2
    xx = 2 * np.arange(w)[None, None, :, None] + code % 2
    argmax = ((yy * (2 * w) + xx) * 64 + np.arange(64)[None, None, None, :]).astype(np.int64)
    want = orc.bottleneck_up(P, "Bottleneck4_0", x, argmax)
    got = net.Bottleneck4_0(dev(x), dev(argmax), training=False, arithmetic="bf16x3").cpu().numpy()
    report_diff("Bottleneck4_0 bf16x3 vs oracle", got, want, exact=False, atol=TOL_LAYER)
    exact = net.Bottleneck4_0(dev(x), dev(argmax), training=False).cpu().numpy()
    report_diff("Bottleneck4_0 default mode still bit-exact", exact, want)
    assert not np.array_equal(got, exact) or got.size < 1024, "the opt-in mode produced the exact kernel's bits: not dispatched?"


def test_layers_without_a_split_kernel_run_exact(enet_c3k19):
    """the mode covers Bottleneck2_0 .. 4_0: every other layer (the first pooling block included) runs the exact kernels"""
    net, P = enet_c3k19
    x16 = np.random.default_rng(3).normal(size=(1, 16, 24, 16)).astype(np.float32)
    want, want_arg = orc.bottleneck_down(P, "Bottleneck1_0", x16)
    x64 = np.random.default_rng(4).normal(size=(1, 12, 20, 64)).astype(np.float32)
    got64 = net.Bottleneck1_1(dev(x64), training=False, arithmetic="bf16x3").cpu().numpy()
    report_diff("Bottleneck1_1 under bf16x3 (no split kernel: exact)", got64, orc.bottleneck(P, "Bottleneck1_1", x64, dil=1, asym=False))


def _forward_checks(net, P, x, tag, tol_logit=1e-4):
    ep = {}
    want = orc.enet_forward(P, x, ep)
    xd = dev(x)
    got = net(xd, training=False, arithmetic="bf16x3").cpu().numpy()
    a1, a2 = net.pooling_argmax()
    report_diff(tag + " argmax1 (bit-identical)", a1.cpu().numpy(), ep["argmax1"])
    report_diff(tag + " argmax2 (bit-identical)", a2.cpu().numpy(), ep["argmax2"])
    report_diff(tag + " logits vs C oracle", got, want, exact=False, atol=tol_logit)
    # a class label may only change where the two best logits of the exact evaluation are closer than twice the error
    flips = np.argwhere(got.argmax(-1) != want.argmax(-1))
    err = float(np.abs(got.astype(np.float64) - want).max())
    for f in flips:
        top2 = np.sort(want[tuple(f)])[-2:]
        assert top2[1] - top2[0] <= 2 * err, "label flip at %s with a logit gap of %g (error %g)" % (tuple(f), top2[1] - top2[0], err)
    return len(flips), err


def test_forward_fixture_sizes_pooling_indices_identical_logits_close(enet_c3k19, enet_c4k6):
    net, P = enet_c3k19
    flips, err = _forward_checks(net, P, frames([0, 1], 64, 128, 3), "64x128")
    assert err < 5e-5
    flips2, err2 = _forward_checks(net, P, frames([0, 1, 2, 3], 256, 512, 3), "C1 256x512")
    assert err2 < 5e-5
    net4, P4 = enet_c4k6
    _forward_checks(net4, P4, frames([5], 72, 136, 4), "C5-shaped 72x136x4")


@pytest.mark.parametrize("measure", ["entropy", "margin", "confidence"])
def test_confidence_and_scores_within_north_star_tolerance(enet_c3k19, measure):
    net, P = enet_c3k19
    for ids, h, w in (([0, 1, 2], 64, 128), ([7], 256, 512)):
        x = frames(ids, h, w, 3)
        want_mean, want_conf, want_label, _ = orc.score_images(P, x, measure)
        scores, extra = net.score(dev(x), measure, return_label=True, return_confidence=True, arithmetic="bf16x3")
        report_diff("%s conf %dx%d" % (measure, h, w), extra["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL_CONF)
        report_diff("%s mean %dx%d" % (measure, h, w), scores.cpu().numpy(), want_mean, exact=False, atol=TOL_SCORE)
        lab = extra["label"].cpu().numpy()
        assert (lab != want_label).mean() <= 1e-4, "label flips: %d" % int((lab != want_label).sum())
        # the ranking-pass form (no outputs: fused ends) gives the same per-image bits as the form with outputs
        assert torch.equal(net.score(dev(x), measure, arithmetic="bf16x3"), scores)


def test_full_resolution_frame(enet_c3k19):
    """BASELINE C2 shape, frame 100 (the frame the exact path is oracle-checked on): per-pixel confidence <= 1e-4, per-image
    score <= 1e-6, pooling indices bit-identical (~25 s of CPU oracle)"""
    net, P = enet_c3k19
    x = frames([100], 1024, 2048, 3)
    ep = {}
    want = orc.enet_forward(P, x, ep)
    want_mean, want_conf, want_label = orc.score_logits(want, "entropy")
    xd = dev(x)
    scores, extra = net.score(xd, "entropy", return_label=True, return_confidence=True, arithmetic="bf16x3")
    a1, a2 = net.pooling_argmax()
    report_diff("1024x2048 argmax1", a1.cpu().numpy(), ep["argmax1"])
    report_diff("1024x2048 argmax2", a2.cpu().numpy(), ep["argmax2"])
    report_diff("1024x2048 confidence", extra["confidence"].cpu().numpy(), want_conf, exact=False, atol=TOL_CONF)
    report_diff("1024x2048 mean", scores.cpu().numpy(), want_mean, exact=False, atol=TOL_SCORE)
    flips = int((extra["label"].cpu().numpy() != want_label).sum())
    assert flips <= 20, "label flips on one full-size frame: %d of 2097152" % flips
    table = pool_score_table("enet", 3, 19, 1024, 2048, "entropy", 0)
    assert abs(scores.cpu().numpy()[0] - table[100]) <= TOL_SCORE


def test_whole_pool_top_128_identical_to_the_exact_path_and_label_census(enet_c3k19):
    """BASELINE configs[1] / [2]: all 2975 full-size frames scored in the opt-in mode; the top-128 selection must be the
    exact path's (the committed, oracle-tied table), with the decision margin at the 128-th boundary far above the
    largest score difference.  Census over the first 480 frames of how many per-pixel class labels differ from the exact
    path's (both evaluated here)."""
    net, _ = enet_c3k19
    table = pool_score_table("enet", 3, 19, 1024, 2048, "entropy", 0)
    pool, bs, k = len(table), 8, 128
    assert pool == 2975
    got = np.empty(pool, dtype=np.float64)
    flips = pixels = 0
    for first in range(0, pool, bs):
        n = min(bs, pool - first)
        x = syn.synth_frames_device(first, n, 1024, 2048, 3)
        if first < 480:
            s, e = net.score(x, "entropy", return_label=True, arithmetic="bf16x3")
            _, e0 = net.score(x, "entropy", return_label=True)
            flips += int((e["label"] != e0["label"]).sum().item())
            pixels += e["label"].numel()
        else:
            s = net.score(x, "entropy", arithmetic="bf16x3")
        got[first:first + n] = s.cpu().numpy()
    dmax = float(np.abs(got - table).max())
    assert dmax <= TOL_SCORE, dmax
    unl = np.arange(pool)
    low, uc = al.finish_ranking(unl, got, pool, unl, k)
    low0, uc0 = al.finish_ranking(unl, table, pool, unl, k)
    srt = np.sort(uc0)
    gap = float(srt[k] - srt[k - 1])
    assert gap > 100 * max(dmax, 1e-9), "decision margin %g at the %d-th boundary vs max score difference %g" % (gap, k, dmax)
    assert sorted(low.tolist()) == sorted(low0.tolist())
    assert int(np.sort(low).sum()) == 190508  # the top-128 checksum of every committed bench record
    frac = flips / float(pixels)
    print("bf16x3 label census: %d of %d pixels (%.2e) over 480 frames differ from the exact path; max |score diff| %.3e; "
          "boundary gap %.3e" % (flips, pixels, frac, dmax, gap))
    assert frac <= 1e-5


def test_rank_confidence_and_argument_errors(enet_c3k19):
    net, P = enet_c3k19
    num, bs, k = 13, 4, 3
    unlabelled = np.array([0, 1, 2, 4, 5, 7, 8, 9, 11, 12])

    def batches():
        for i in range(0, num, bs):
            ids = np.arange(i, min(i + bs, num))
            yield frames(ids, 64, 64, 3), ids

    low, uc = al.rank_confidence(net, batches(), num, unlabelled, k, measure="entropy", arithmetic="bf16x3")
    want_scores = np.concatenate([orc.score_images(P, frames([i], 64, 64, 3))[0] for i in range(num)])
    want_low, want_uc = orc.rank_lowest(want_scores, unlabelled, k)
    assert set(low.tolist()) == set(want_low.tolist())
    report_diff("unlabelled_confidence", uc, want_uc, exact=False, atol=TOL_SCORE)
    x = dev(frames([0], 64, 64, 3))
    with pytest.raises(ValueError):
        net.score(x, arithmetic="fp8")
    with pytest.raises(ValueError):
        net(x, training=False, arithmetic="tf32")
    ws = torch.empty(int(_lib.lib().ssal_enet_workspace_bytes(net._sync_handle(), 1, 64, 64)), dtype=torch.uint8, device="cuda")
    sc = torch.empty(1, dtype=torch.float64, device="cuda")
    rc = _lib.lib().ssal_enet_score_nhwc_arith(net._sync_handle(), _lib.dev_ptr(x), 0, 1, 64, 64, 0, 0.0, 7, _lib.dev_ptr(sc), None,
                                                None, None, _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr())
    assert rc == _lib.SSAL_EINVAL and b"arithmetic" in _lib.lib().ssal_last_error()
    # arithmetic = SSAL_ARITH_F32 through the _arith entry points IS the default path
    rc = _lib.lib().ssal_enet_score_nhwc_arith(net._sync_handle(), _lib.dev_ptr(x), 0, 1, 64, 64, 0, 0.0, 0, _lib.dev_ptr(sc), None,
                                                None, None, _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr())
    assert rc == 0 and torch.equal(sc, net.score(x, "entropy"))
