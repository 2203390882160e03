"""CPU tests of the ICNet row (no GPU): the C restatement (oracle/icnet_oracle.py) against the committed golden
fixture, against the independent torch-CPU restatement, against naive numpy statements of the TF ops ICNET_SPEC.md
names; the host-side model surface (models.ICNet) and the C-ABI tensor inventory.

PARITY STATUS: unpinned AND undefined -- the reference's models/icnet/icnet.py:1-7 is an empty class; nothing here
is anchored in reference behaviour (ICNET_SPEC.md section 7)."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import semanticsegmentationactivelearning_amd as ssal
from helpers import frames, report_diff
from oracle import enet_oracle as orc
from oracle import icnet_oracle as ico
from oracle import torch_restatement as tr
from semanticsegmentationactivelearning_amd import _lib, synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sha(arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def icnet19():
    net = ssal.ICNet(19)
    net.build((None, None, None, 3))
    syn.randomize_icnet(net, seed=0)
    return net, syn.icnet_params_dict(net)


def test_oracle_reproduces_golden_fixture(icnet19):
    net, P = icnet19
    g = np.load(os.path.join(GOLDEN, "icnet_c3k19_64x128.npz"))
    assert str(g["weights_sha256"]) == _sha([P[k] for k in sorted(P)]), "synthetic weight recipe drifted"
    x = frames(list(g["frame_ids"]), 64, 128, 3)
    assert str(g["frames_sha256"]) == _sha([x])
    ep = {}
    logits = ico.icnet_forward(P, x, ep)
    report_diff("1/4-resolution logits", ep["conv6_cls"], g["logits_quarter"])
    report_diff("label", logits.argmax(-1).astype(np.uint8), g["label"])
    report_diff("conv5_3_sum slice", ep["conv5_3_sum"][0, :, :, :8], g["conv5_3_sum_slice"])
    assert float(ep["sub24_sum"].astype(np.float64).sum()) == float(g["sub24_sum_checksum"][0])
    for m in ("entropy", "margin", "confidence"):
        mean, conf, label = orc.score_logits(logits, m)
        report_diff(m + " mean", mean, g["mean_" + m], exact=False, atol=1e-12)
        report_diff(m + " conf", conf[0], g["conf_" + m])


def test_oracle_agrees_with_torch_restatement(icnet19):
    """two independent restatements of ICNET_SPEC.md (fixed-order C with folded batch-norm vs stock torch ops with
    batch-norm as TF writes it): every shared endpoint and the logits within 1e-4"""
    net, P = icnet19
    x = frames([5, 6], 64, 96, 3)
    ea, eb = {}, {}
    la, lb = ico.icnet_forward(P, x, ea), tr.icnet_forward(P, x, eb)
    for k in eb:
        report_diff(k, ea[k], eb[k], exact=False, atol=1e-4 * max(1.0, float(np.abs(ea[k]).max()) / 30.0))
    report_diff("logits", la, lb, exact=False, atol=1e-4)
    for m in ("entropy", "margin", "confidence"):
        ma, ca, ba = orc.score_logits(la, m)
        mb, cb, bb = tr.score_logits(lb, m)
        assert np.abs(ma - mb).max() < 1e-5
        assert (ba != bb).mean() < 1e-3  # labels differ only where two logits tie within rounding


def test_max_pool_3x3_s2_same_against_naive():
    rng = np.random.default_rng(0)
    for h, w in ((8, 10), (7, 9), (2, 2), (1, 5)):
        x = rng.normal(size=(2, h, w, 3)).astype(np.float32)
        ho, wo = (h + 1) // 2, (w + 1) // 2
        th, tw = max((ho - 1) * 2 + 3 - h, 0), max((wo - 1) * 2 + 3 - w, 0)
        xp = np.full((2, h + th, w + tw, 3), -np.inf, np.float32)
        xp[:, th // 2: th // 2 + h, tw // 2: tw // 2 + w] = x
        want = np.stack([[xp[:, 2 * oy: 2 * oy + 3, 2 * ox: 2 * ox + 3].max(axis=(1, 2)) for ox in range(wo)]
                         for oy in range(ho)]).transpose(2, 0, 1, 3)
        report_diff("maxpool %dx%d" % (h, w), ico.maxpool3x3_s2(x), want)
        got_t = tr.max_pool_3x3_s2_same(tr._t(x).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
        report_diff("torch maxpool", got_t, want)


def test_resize_bilinear_legacy_against_literal_statement():
    """tf.image.resize_bilinear, TF-1.13 defaults (inference.py:96-99): src = dst * in/out, no half-pixel offset"""
    rng = np.random.default_rng(1)
    for (h, w), (oh, ow) in (((4, 6), (8, 12)), ((8, 8), (4, 4)), ((3, 5), (12, 20)), ((5, 7), (5, 7)), ((2, 3), (7, 4))):
        x = rng.normal(size=(2, h, w, 3)).astype(np.float32)
        want = np.empty((2, oh, ow, 3), np.float32)
        hs, ws = np.float32(h) / np.float32(oh), np.float32(w) / np.float32(ow)
        for oy in range(oh):
            fy = np.float32(oy) * hs
            y0 = int(np.floor(fy)); y1 = min(y0 + 1, h - 1); ly = np.float32(fy - np.float32(y0))
            for ox in range(ow):
                fx = np.float32(ox) * ws
                x0 = int(np.floor(fx)); x1 = min(x0 + 1, w - 1); lx = np.float32(fx - np.float32(x0))
                top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
                bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
                want[:, oy, ox] = top + (bot - top) * ly
        report_diff("resize", ico.resize_bilinear(x, oh, ow), want)
        got_t = tr.resize_bilinear_legacy(tr._t(x).permute(0, 3, 1, 2), oh, ow).permute(0, 2, 3, 1).numpy()
        report_diff("torch resize", got_t, want, exact=False, atol=1e-6)
    # the exact factors ICNET_SPEC relies on: 1/2 picks x[2y][2x]; 2x leaves even positions untouched
    x = rng.normal(size=(1, 8, 8, 2)).astype(np.float32)
    assert np.array_equal(ico.resize_bilinear(x, 4, 4), x[:, ::2, ::2])
    assert np.array_equal(ico.resize_bilinear(x, 16, 16)[:, ::2, ::2], x)


def test_pyramid_pooling_against_numpy():
    rng = np.random.default_rng(2)
    for h, w in ((32, 64), (7, 5), (2, 4), (1, 1)):
        x = rng.normal(size=(2, h, w, 4)).astype(np.float32)
        for b in ico.PPM_BINS:
            got = ico.adaptive_avg_pool(x, b)
            for i in range(b):
                for j in range(b):
                    y0, y1 = (i * h) // b, -(-((i + 1) * h) // b)
                    x0, x1 = (j * w) // b, -(-((j + 1) * w) // b)
                    want = x[:, y0:y1, x0:x1].astype(np.float64).mean(axis=(1, 2))
                    assert np.abs(got[:, i, j] - want).max() < 1e-5
        acc = x.astype(np.float64)
        for b in ico.PPM_BINS:
            acc = acc + ico.resize_bilinear(ico.adaptive_avg_pool(x, b), h, w)
        assert np.abs(ico.pyramid_pooling(x) - acc).max() < 1e-5


def test_affine_add_relu():
    rng = np.random.default_rng(3)
    x, r = rng.normal(size=(2, 3, 3, 8)).astype(np.float32), rng.normal(size=(2, 3, 3, 8)).astype(np.float32)
    s, t = rng.uniform(0.5, 1.5, 8).astype(np.float32), rng.normal(size=8).astype(np.float32)
    want = np.maximum((x.astype(np.float64) * s + t) + r, 0)
    assert np.abs(ico.affine_add_relu(x, s, t, r, True) - want).max() < 1e-6
    assert np.array_equal(ico.affine_add_relu(x, None, t, None, False), x + t)


# ---- host-side model surface ---------------------------------------------------------------------
def test_icnet_parameter_inventory_and_macs(icnet19):
    net, P = icnet19
    assert set(P) == set(ico.param_shapes(3, 19))
    for k, shp in ico.param_shapes(3, 19).items():
        assert P[k].shape == shp, k
    assert sum(v.size for v in P.values()) == 6728531          # ICNET_SPEC.md section 5
    assert ico.macs_per_image(1024, 2048) == 28122808320       # 28.12 GMAC per 1024x2048x3 image
    assert len(net.layers) == 64 and net.layers[-1].name == "conv6_cls"  # 63 conv+BN layers + the classifier
    # Keras variable order: trainable first (kernel, gamma, beta), moving statistics last
    assert [v.name for v in net.conv1_1_3x3_s2.variables] == [
        "conv1_1_3x3_s2/Kernel", "conv1_1_3x3_s2/BatchNorm/Gamma", "conv1_1_3x3_s2/BatchNorm/Beta",
        "conv1_1_3x3_s2/BatchNorm/Mean", "conv1_1_3x3_s2/BatchNorm/Variance"]
    assert [v.name for v in net.conv6_cls.variables] == ["conv6_cls/Kernel", "conv6_cls/Bias"]
    assert net.conv_sub4.kernel.shape == (3, 3, 256, 128) and net.conv5_3_1x1_increase.kernel.shape == (1, 1, 256, 1024)


def test_icnet_handle_tensor_inventory_matches_python_model(icnet19):
    net, P = icnet19
    lib = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(lib.ssal_icnet_create(3, 19, ctypes.byref(h)))
    seen = {}
    for i in range(lib.ssal_icnet_num_tensors(h)):
        name, nd, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        _lib.check(lib.ssal_icnet_tensor_info(h, i, ctypes.byref(name), ctypes.byref(nd), dims))
        seen[name.value.decode()] = tuple(dims[: nd.value])
    assert set(seen) == set(P)
    for k, v in P.items():
        assert seen[k] == v.shape, k
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_icnet_set_tensor(h, b"conv6_cls.bias", np.zeros(3, np.float32).ctypes.data_as(ctypes.c_void_p), 3))
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_icnet_set_tensor(h, b"nope.kernel", np.zeros(3, np.float32).ctypes.data_as(ctypes.c_void_p), 3))
    assert lib.ssal_icnet_workspace_bytes(h, 1, 64, 64) == -1  # not committed
    names = []
    for i in range(lib.ssal_icnet_num_endpoints(h)):
        p = ctypes.c_char_p()
        _lib.check(lib.ssal_icnet_endpoint_name(h, i, ctypes.byref(p)))
        names.append(p.value.decode())
    assert len(names) == 67 and "sub24_sum" in names and "conv5_4_interp" not in names
    off, dims = ctypes.c_int64(), (ctypes.c_int64 * 4)()
    _lib.check(lib.ssal_icnet_endpoint_info(h, b"conv3_1", 2, 64, 128, ctypes.byref(off), dims))
    assert tuple(dims) == (2, 4, 8, 256) and off.value % 256 == 0
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_icnet_create(2, 19, ctypes.byref(ctypes.c_void_p())))
    lib.ssal_icnet_destroy(h)


def test_icnet_rejects_training_and_has_no_cpu_fallback():
    import torch
    net = ssal.ICNet(19)
    with pytest.raises(NotImplementedError):
        net(np.zeros((1, 32, 32, 3), np.float32), training=True)
    with pytest.raises(NotImplementedError):
        net.score(np.zeros((1, 32, 32, 3), np.float32), measure="bald")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net(np.zeros((1, 32, 32, 3), np.float32), training=False)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net.score(np.zeros((1, 32, 32, 3), np.float32))


def test_icnet_name_keyed_weight_copy(icnet19):
    a, _ = icnet19
    b = ssal.ICNet(19)
    b.build((None, None, None, 3))
    assert b.assign_named({"ICNet/%s:0" % v.name: v.numpy() for v in a.variables}, strict=True) == len(a.variables)
    pa, pb = syn.icnet_params_dict(a), syn.icnet_params_dict(b)
    assert all((pa[k] == pb[k]).all() for k in pa)
