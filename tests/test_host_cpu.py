"""CPU tests of the host side: the C-ABI library loads and exports every symbol the header
declares, the models.ENet mirror keeps the reference's API surface, the ranking tail and the
synthetic generators behave.  No kernel is launched here."""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import _lib, active_learning as al, synthetic as syn
from semanticsegmentationactivelearning_amd.models.enet import enet_modules as mod

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ssal_enet.h")


def header_functions():
    names = set()
    for hdr in (HEADER, os.path.join(os.path.dirname(HEADER), "ssal_icnet.h")):
        src = open(hdr).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(ssal_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    names = header_functions()
    assert len(names) >= 20
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with include/ssal_enet.h"
    lib = _lib.lib()  # loads libssal_hip.so; getattr fails if a symbol is missing
    for n in names:
        assert hasattr(lib, n)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (ssal_[a-z0-9_]+)", out))
    assert exported == set(names), exported ^ set(names)
    assert b"gfx950" in lib.ssal_version()


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"amdgcn-amd-amdhsa" in blob


def test_handle_tensor_inventory_matches_python_model(enet_c3k19):
    """no GPU needed: create/destroy a handle and compare its tensor list with the Python mirror"""
    import ctypes
    net, P = enet_c3k19
    lib = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(lib.ssal_enet_create(3, 19, ctypes.byref(h)))
    n = lib.ssal_enet_num_tensors(h)
    seen = {}
    for i in range(n):
        name, nd, dims = ctypes.c_char_p(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        _lib.check(lib.ssal_enet_tensor_info(h, i, ctypes.byref(name), ctypes.byref(nd), dims))
        seen[name.value.decode()] = tuple(dims[: nd.value])
    assert set(seen) == set(P)
    for k, v in P.items():
        assert seen[k] == v.shape, k
    # error paths of the staging API
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_enet_set_tensor(h, b"Nope.kernel", P["Final.kernel"].ctypes.data_as(ctypes.c_void_p), 3))
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_enet_set_tensor(h, b"Final.kernel", P["Final.kernel"].ctypes.data_as(ctypes.c_void_p), 3))
    assert lib.ssal_enet_workspace_bytes(h, 1, 64, 64) == -1  # not committed
    _lib.check(lib.ssal_enet_destroy(h))
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_enet_create(5, 19, ctypes.byref(h)))
    with pytest.raises(ValueError):
        _lib.check(lib.ssal_enet_create(3, 64, ctypes.byref(h)))


def test_enet_constructor_contract():
    with pytest.raises(ValueError, match="drop_rates"):
        ssal.ENet(19, drop_rates=[0.1, 0.1])
    net = ssal.ENet(19)
    names = [l.name for l in net.layers]
    assert names[0] == "Initial" and names[-1] == "Final" and len(names) == 29
    assert isinstance(net.Bottleneck1_0, mod.BottleneckDownsample)
    assert isinstance(net.Bottleneck4_0, mod.BottleneckUpsample)
    assert net.Bottleneck2_3.asymmetric and net.Bottleneck3_7.asymmetric
    assert net.Bottleneck2_8.dilation_rate == (16, 16) and net.Bottleneck3_2.dilation_rate == (2, 2)
    assert net.endpoint_outputs == []


@pytest.mark.parametrize("classes,c_in,count", [(19, 3, 377007), (6, 4, 375216)])
def test_parameter_count_and_layouts(classes, c_in, count):
    """377 007 parameters (C2) / 375 216 (C5)"""
    net = ssal.ENet(classes)
    net.build((None, None, None, c_in))
    total = sum(int(np.prod(v.shape)) for v in net.variables)  # kernels, alphas, BN gamma/beta + moving stats
    assert total == count
    assert net.Initial.kernel.shape == (3, 3, c_in, 16 - c_in)
    assert net.Bottleneck1_0.proj_kernel.shape == (2, 2, 16, 8)
    assert net.Bottleneck2_3.conv_kernel[0].shape == (5, 1, 32, 32)
    assert net.Bottleneck2_3.conv_kernel[1].shape == (1, 5, 32, 32)
    assert net.Bottleneck4_0.conv_kernel.shape == (3, 3, 16, 32)  # HW-O-I
    assert net.Bottleneck4_0.res_kernel.shape == (1, 1, 128, 64)
    assert net.Bottleneck5_0.exp_kernel.shape == (1, 1, 8, 16)
    assert net.Final.kernel.shape == (3, 3, classes, 16)
    assert len(net.Bottleneck1_1.variables) == 18


# Keras (TF 1.13) `Layer.variables` == `Layer.weights` == trainable_weights + non_trainable_weights, each group in
# add_weight order.  The lists below are written out from the reference's `trainable=` flags
# (models/enet/enet_modules.py:139-187 Initial, 366-523 Bottleneck, 1070-1214 BottleneckUpsample, 1349-1356 Final):
# the moving batch-norm Mean / Variance (trainable=False) come last.
_BN_T = ["BatchNorm/Gamma", "BatchNorm/Beta"]
_BN_N = ["BatchNorm/Mean", "BatchNorm/Variance"]
KERAS_ORDER = {
    "Initial": ["Convolution/Kernel", "Convolution/BatchNorm/Gamma", "Convolution/BatchNorm/Beta", "Residual/Alpha",
                "Convolution/BatchNorm/Mean", "Convolution/BatchNorm/Variance"],
    "Bottleneck1_1": ["Projection/Kernel", "Projection/Alpha"] + ["Projection/" + t for t in _BN_T]
                     + ["Convolution/Kernel", "Convolution/Alpha"] + ["Convolution/" + t for t in _BN_T]
                     + ["Expansion/Kernel"] + ["Expansion/" + t for t in _BN_T] + ["Residual/Alpha"]
                     + ["Projection/" + t for t in _BN_N] + ["Convolution/" + t for t in _BN_N]
                     + ["Expansion/" + t for t in _BN_N],
    "Bottleneck2_3": ["Projection/Kernel", "Projection/Alpha"] + ["Projection/" + t for t in _BN_T]
                     + ["Convolution/KernelCol", "Convolution/KernelRow", "Convolution/Alpha"]
                     + ["Convolution/" + t for t in _BN_T]
                     + ["Expansion/Kernel"] + ["Expansion/" + t for t in _BN_T] + ["Residual/Alpha"]
                     + ["Projection/" + t for t in _BN_N] + ["Convolution/" + t for t in _BN_N]
                     + ["Expansion/" + t for t in _BN_N],
    "Bottleneck4_0": ["Projection/Kernel", "Projection/Alpha"] + ["Projection/" + t for t in _BN_T]
                     + ["Convolution/Kernel", "Convolution/Alpha"] + ["Convolution/" + t for t in _BN_T]
                     + ["Expansion/Kernel"] + ["Expansion/" + t for t in _BN_T]
                     + ["Residual/Kernel", "Residual/Alpha"]
                     + ["Projection/" + t for t in _BN_N] + ["Convolution/" + t for t in _BN_N]
                     + ["Expansion/" + t for t in _BN_N],
    "Final": ["Kernel"],
}


@pytest.mark.parametrize("layer", sorted(KERAS_ORDER))
def test_layer_variables_follow_the_keras_order(layer):
    net = ssal.ENet(19)
    net.build((None, None, None, 3))
    L = getattr(net, layer)
    leafs = [v.name.split("/", 1)[1] for v in L.variables]
    assert leafs == KERAS_ORDER[layer]
    assert [v.name for v in L.weights] == [v.name for v in L.variables]
    assert all(v.trainable for v in L.trainable_weights) and not any(v.trainable for v in L.non_trainable_weights)
    # creation order stays available (the seeded synthetic recipe draws in it)
    assert sorted(v.name for v in L.creation_order_variables) == sorted(v.name for v in L.variables)
    if layer == "Initial":
        assert [v.name.split("/", 1)[1] for v in L.creation_order_variables][:3] == \
            ["Convolution/Kernel", "Convolution/BatchNorm/Mean", "Convolution/BatchNorm/Variance"]


def test_positional_copy_from_a_tf_ordered_list():
    """A list laid out like TF's `Initial.variables` ([kernel, gamma, beta, alpha, mean, variance]) copied
    positionally (active_learning.py:475-482) must land gamma in gamma and variance in variance."""
    net = ssal.ENet(19)
    net.build((None, None, None, 3))
    rng = np.random.default_rng(0)
    src = {"kernel": rng.normal(size=(3, 3, 3, 13)), "gamma": rng.uniform(0.8, 1.2, 16), "beta": rng.normal(size=16),
           "alpha": rng.uniform(0.1, 0.4, 16), "mean": rng.normal(size=16), "variance": rng.uniform(0.5, 1.5, 16)}
    tf_list = [src[k] for k in ("kernel", "gamma", "beta", "alpha", "mean", "variance")]
    for dst, val in zip(net.Initial.variables, tf_list):
        dst.assign(val)
    for k, v in src.items():
        assert np.array_equal(getattr(net.Initial, k).numpy(), v.astype(np.float32)), k
    assert (net.Initial.variance.numpy() > 0).all()


def test_name_keyed_weight_copy():
    a, b = ssal.ENet(19), ssal.ENet(19)
    a.build((None, None, None, 3))
    b.build((None, None, None, 3))
    syn.randomize_enet(a, seed=5)
    named = {"ENet/%s:0" % v.name: v.numpy().copy() for v in a.variables}  # TF-style names
    assert b.assign_named(named) == len(a.variables)
    pa, pb = syn.enet_params_dict(a), syn.enet_params_dict(b)
    assert all((pa[k] == pb[k]).all() for k in pa)
    with pytest.raises(KeyError):
        b.assign_named({"ENet/Initial/NoSuchThing:0": np.zeros(3)})
    del named["ENet/Final/Kernel:0"]
    with pytest.raises(KeyError):
        b.assign_named(named, strict=True)


def test_positional_weight_copy_between_models():
    """active_learning.py:475-482: val_net.layers[i].variables[j] <- train_net ..."""
    a, b = ssal.ENet(19), ssal.ENet(19)
    a.build((None, None, None, 3))
    b.build((None, None, None, 3))
    syn.randomize_enet(a, seed=3)
    for la, lb in zip(a.layers, b.layers):
        for va, vb in zip(la.variables, lb.variables):
            vb.assign(va)
    pa, pb = syn.enet_params_dict(a), syn.enet_params_dict(b)
    assert all((pa[k] == pb[k]).all() for k in pa)
    with pytest.raises(ValueError):
        b.Final.kernel.assign(np.zeros((3, 3, 5, 16), np.float32))


def test_training_mode_is_rejected_without_touching_the_gpu():
    net = ssal.ENet(19)
    with pytest.raises(NotImplementedError):
        net(np.zeros((1, 8, 8, 3), np.float32), training=True)
    with pytest.raises(NotImplementedError):
        net.score(np.zeros((1, 8, 8, 3), np.float32), measure="bald")


def test_no_cpu_fallback_when_no_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = ssal.ENet(19)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(np.zeros((1, 8, 8, 3), np.float32), training=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        al.score_logits(np.zeros((1, 8, 8, 19), np.float32))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semanticsegmentationactivelearning_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libenet_oracle" not in src, f


def test_select_lowest_and_reference_defect():
    conf = np.array([0.5, 0.1, 0.9, 0.3, 0.2], np.float32)
    assert set(al.select_lowest(conf, 2).tolist()) == {1, 4}
    assert set(al.select_lowest(conf, 5).tolist()) == {0, 1, 2, 3, 4}  # reference raises here (kth == len)
    assert set(al.select_lowest(conf, 50).tolist()) == {0, 1, 2, 3, 4}
    assert al.select_lowest(conf, 0).size == 0


def test_finish_ranking_scatter_and_float32_rounding():
    idx = np.array([3, 0, -1, 2, 1, -1])
    sc = np.array([0.30000000001, 0.9, np.inf, 0.1, 0.5, np.inf])
    low, uc = al.finish_ranking(idx, sc, 4, unlabelled=[0, 2, 3], selection_size=1)
    assert low.tolist() == [2]
    assert uc.dtype == np.float32 and uc.tolist() == [np.float32(0.9), np.float32(0.1), np.float32(0.30000000001)]


def test_shard_positions_cover_pool_once():
    n, world = 2975, 8
    shards = [al.shard_positions(n, r, world) for r in range(world)]
    assert all(len(s) == 372 for s in shards)
    allpos = np.concatenate(shards)
    assert sorted(allpos[allpos >= 0].tolist()) == list(range(n))
    assert (allpos < 0).sum() == 8 * 372 - n


def test_scoring_config_reads_reference_json_keys():
    params = json.loads("""{"batch_size": 8, "network": {"model": "ENet", "input": {"height": 432, "width": 648}},
        "active_learning": {"selection_size": 50, "measure": "entropy", "threshold": 0.95}}""")
    cfg = al.ScoringConfig.from_params(params)
    assert (cfg.measure, cfg.selection_size, cfg.threshold, cfg.batch_size, cfg.height, cfg.width) == \
        ("entropy", 50, 0.95, 8, 432, 648)
    params["active_learning"]["measure"] = "bald"
    with pytest.raises(NotImplementedError):
        al.ScoringConfig.from_params(params)
    assert al.EPSILON == np.finfo(np.float32).tiny


def test_synthetic_frames_are_deterministic_and_structured():
    a = syn.synth_frame_u8(7, 64, 128, 3)
    b = syn.synth_frame_u8(7, 64, 128, 3)
    c = syn.synth_frame_u8(8, 64, 128, 3)
    assert a.dtype == np.uint8 and a.shape == (64, 128, 3)
    assert (a == b).all() and (a != c).any()
    assert abs(float(a.mean()) - float(c.mean())) > 0.5  # per-frame brightness differs
    blk = a[:8, :8, 0].astype(int)
    assert blk.max() - blk.min() <= 33  # coarse 8x8 block + fine noise in [-16,16]
    x = syn.u8_to_f32(a)
    assert x.dtype == np.float32 and x.max() <= 1.0 and (x == a.astype(np.float32) * np.float32(1 / 255)).all()


def test_glorot_matches_tf_fan_rule():
    init = mod.glorot_uniform(seed=0)
    w = init((3, 3, 16, 32))
    lim = np.sqrt(6.0 / (9 * 16 + 9 * 32))
    assert w.dtype == np.float32 and np.abs(w).max() <= lim and np.abs(w).max() > 0.9 * lim
    a = init([32])  # conv_alpha uses the kernel initializer on a 1-D shape (enet_modules.py:442-449)
    assert np.abs(a).max() <= np.sqrt(6.0 / 64)


def test_bench_score_digest_against_table(tmp_path, monkeypatch):
    """bench.score_digest: SHA-256 of the float64 scores in frame order vs the same frames of the committed table; sentinel
    (-1) entries and duplicates of a wrapped batch list are dropped; a differing bit is a mismatch; no table -> None"""
    import bench
    table = np.linspace(0.1, 0.9, 50)
    key = bench.table_key("enet", 3, 19, 1024, 2048, "entropy", 0)
    assert key == "enet_c3k19_1024x2048_entropy_seed0"
    path = tmp_path / "pool_scores.npz"
    np.savez(path, **{key: table})
    monkeypatch.setattr(bench, "SCORE_TABLE", str(path))
    idx = np.array([7, 3, -1, 12, 3, -1])
    sc = np.where(idx >= 0, table[np.maximum(idx, 0)], np.inf)
    d = bench.score_digest(idx, sc, key)
    assert d["match"] is True and d["frames"] == 3 and d["sha256"] == d["expected_sha256"]
    sc2 = sc.copy()
    sc2[0] = np.nextafter(sc2[0], 1.0)
    d2 = bench.score_digest(idx, sc2, key)
    assert d2["match"] is False and 0 < d2["max_abs_diff"] < 1e-15
    assert bench.score_digest(idx, sc, "no_such_key")["match"] is None
    assert bench.score_digest(np.array([60]), np.array([0.5]), key)["match"] is None  # frame beyond the table


def test_committed_pool_score_table_has_every_bench_config():
    import bench
    with np.load(bench.SCORE_TABLE) as z:
        assert z[bench.table_key("enet", 3, 19, 1024, 2048, "entropy", 0)].shape == (bench.POOL,)
        for key in (bench.table_key("icnet", 3, 19, 1024, 2048, "margin", 0), bench.table_key("enet", 4, 6, 1024, 2048, "entropy", 1)):
            assert z[key].shape[0] >= 8 * (24 + 2) and z[key].dtype == np.float64 and np.isfinite(z[key]).all()


def test_bench_watchdog_exits_nonzero_when_a_block_hangs():
    """bench.Watchdog: a rendezvous / barrier that never returns ends the process with exit code 3 instead of hanging"""
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog(0.5, 'test block'):\n    time.sleep(30)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=25)
    assert r.returncode == 3 and b"TIMEOUT" in r.stderr
    ok = ("import sys; sys.path.insert(0, %r); import bench\n"
          "with bench.Watchdog(5, 'fast block'):\n    pass\nprint('done')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", ok], capture_output=True, timeout=25)
    assert r.returncode == 0 and b"done" in r.stdout


def test_bench_scaling_default_and_common_step_count():
    """strong is the default with WORLD_SIZE > 1; every rank derives the same step count from ceil(POOL / world) (ADVICE
    r02: per-rank ceil(len / batch) can differ between the long and the short shards, e.g. batch 7 on 8 ranks)"""
    import bench
    a = bench.parse([])
    assert a.scaling is None and a.steps == 372 and a.dist_timeout > 0
    for world, bs in ((8, 8), (8, 7), (2, 8), (3, 5)):
        per = (bench.POOL + world - 1) // world
        steps = (per + bs - 1) // bs
        for rank in range(world):
            mine = len(range(rank, bench.POOL, world))
            assert mine <= per <= steps * bs


def test_bench_kernel_roofline_rows_and_committed_pmc():
    """bench.kernel_roofline: bound from the arithmetic intensity against the fp32 ridge, achieved / frac per launch, PMC
    traffic per launch from the newest committed pass (profiles/r05_pmc, with the opt-in kernels' own pass merged in); packed-vector
    kernels are labelled as such"""
    import bench
    rnd, pmc = bench.load_pmc("enet")
    assert rnd == "r05" and "k_bottleneck_mfma<32>" in pmc and "k_final_score<fused 5_1>" in pmc and "k_initial_down16" in pmc
    assert "k_bottleneck_bf16x3" in pmc and "k_bottleneck_asym_bf16x3" in pmc
    d = {"launches": 30, "ms": 30 * 0.114, "flops": 30 * 9.127e9, "bytes": 30 * 268.5e6}
    r = bench.kernel_roofline("k_bottleneck_mfma<32>", d, 3, pmc["k_bottleneck_mfma<32>"])
    assert r["bound"] == "mfma" and r["pipe"] == "mfma" and r["launches_per_batch"] == 10 and abs(r["avg_us"] - 114.0) < 1e-6
    assert abs(r["frac"] - 9.127e9 / 114e-6 / 157.3e12) < 1e-3 and 1.5 < r["traffic_over_algorithmic"] < 1.7
    h = bench.kernel_roofline("k_bottleneck16<32,64,16>", {"launches": 6, "ms": 6 * 0.136, "flops": 6 * 9.13e9, "bytes": 6 * 537e6}, 1, None)
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["traffic"] is None and abs(h["frac"] - 537e6 / 136e-6 / 8e12) < 1e-3
    v = bench.kernel_roofline("k_final_score<fused 5_1>", {"launches": 1, "ms": 0.355, "flops": 23.9e9, "bytes": 268e6}, 1, None)
    assert v["bound"] == "mfma" and v["pipe"].startswith("valu")
    z = bench.kernel_roofline("k_reduce_mean", {"launches": 1, "ms": 0.003, "flops": 0.0, "bytes": 0.0}, 1, None)
    assert z["bound"] == "latency" and z["frac"] is None


def test_bench_line_is_compact_and_keeps_every_leg():
    """the driver keeps a bounded tail of stdout: the printed line must stay under bench.LINE_LIMIT bytes and still carry
    the contract fields, the dominant kernel's roofline, one compact row per kernel and BOTH secondary legs (round 3's
    16 KB line lost secondary.c4); the full figures go to the detail file the line names"""
    import json
    import bench
    full = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", "r03_bench_enet_full_pool.json")))
    assert len(json.dumps(full)) > 15000  # the round-3 line, as the driver received it
    text = bench.compact_line(full, os.path.join(bench.ROOT, "gpurun_out", "bench_detail.json"))
    assert len(text) <= bench.LINE_LIMIT
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "score_digest"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in line["roofline"], k
    assert line["secondary"]["c4"]["value"] > 0 and line["secondary"]["c5"]["value"] > 0
    assert "undefined" in line["secondary"]["c4"]["parity"]
    assert set(line["roofline_all"]) == set(full["roofline_all"])
    assert line["detail_file"] == os.path.join("gpurun_out", "bench_detail.json")


def _import_env(env_extra, code):
    """run `code` in a fresh interpreter (the runtime-environment defaults are applied once, at import)"""
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY", "SSAL_BENCH_INJECTED_HWQ")}
    env.update(env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-W", "always", "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().splitlines()[-1], out.stderr


def test_package_import_defaults_the_hip_runtime_environment():
    """VERDICT r04 item 5a / ADVICE: a ranking job that only imports the package (not bench.py) must get the hardware-queue
    count the bench measures.  Unset or EMPTY -> 8 (an empty value behaves like 2: -11 %); an explicit value is kept, and a
    too small one is reported once by `warn_if_few_hw_queues` (what rank_confidence calls under a process group)."""
    code = ("import os, json; from semanticsegmentationactivelearning_amd import _lib; "
            "w = [_lib.warn_if_few_hw_queues(), _lib.warn_if_few_hw_queues()]; "
            "print(json.dumps([os.environ.get('GPU_MAX_HW_QUEUES'), os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), "
            "sorted(_lib.ENV_INJECTED), w]))")
    line, _ = _import_env({}, code)
    assert json.loads(line) == ["8", "0", ["GPU_MAX_HW_QUEUES", "HSA_ENABLE_IPC_MODE_LEGACY"], [False, False]]
    line, _ = _import_env({"GPU_MAX_HW_QUEUES": "", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, code)
    assert json.loads(line) == ["8", "0", ["GPU_MAX_HW_QUEUES"], [False, False]]
    line, err = _import_env({"GPU_MAX_HW_QUEUES": "2"}, code)
    assert json.loads(line) == ["2", "0", ["HSA_ENABLE_IPC_MODE_LEGACY"], [True, False]]  # warned exactly once
    assert err.count("GPU_MAX_HW_QUEUES='2'") == 1 and "RuntimeWarning" in err


def test_last_call_state_is_per_thread_and_never_another_devices_handle():
    """`_ws` / `_last_dims` / `_last_call` are ONE record per thread, and `_handle` has no any-device fallback"""
    import threading

    class Dummy(_lib.DeviceState):
        pass

    class FakeWs:
        def __init__(self, idx):
            class D:
                index = idx
            self.device = D()

    d = Dummy()
    d._init_device_state()
    d._handles = {0: ["h0", None], 1: ["h1", None]}
    seen = {}

    def worker(tag, dev, dims):
        d._note_call(FakeWs(dev), dims, "score")
        time.sleep(0.05)
        seen[tag] = (d._last_dims, d._last_call, d._handle, d._ws.device.index)

    ts = [threading.Thread(target=worker, args=("a", 0, (1, 32, 32))), threading.Thread(target=worker, args=("b", 1, (2, 64, 64)))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert seen == {"a": ((1, 32, 32), "score", "h0", 0), "b": ((2, 64, 64), "score", "h1", 1)}
    assert d._last_dims is None and d._ws is None  # this thread has made no call
    d._handles = {3: ["h3", None]}
    d._note_call(FakeWs(5), (1, 8, 8), "forward")
    assert d._handle is None  # device 5 has no handle: None, not device 3's


def test_bench_bf16x3_rows_and_compact_leg():
    """the opt-in leg is priced on the bf16 pipe (six products per fp32 product against the dense bf16 peak) and stays a
    SECONDARY leg of the line: its own dtype string, its score check within 1e-6 instead of the bit digest, its own roofline row"""
    import bench
    row = bench.kernel_roofline("k_bottleneck_bf16x3", {"launches": 12, "ms": 12 * 0.088, "flops": 12 * 9.127e9, "bytes": 12 * 268.5e6}, 1,
                                {"hbm_bytes_per_launch": 416.6e6})
    assert row["bound"] == "hbm" and row["pipe"].startswith("mfma bf16") and abs(row["flops"] - 6 * 9.127e9) < 1e6
    assert abs(row["frac"] - 268.5e6 / 88e-6 / 8e12) < 1e-3 and abs(row["traffic_over_algorithmic"] - 416.6 / 268.5) < 1e-3
    exact = bench.kernel_roofline("k_bottleneck_mfma<32>", {"launches": 10, "ms": 1.10, "flops": 10 * 9.127e9, "bytes": 10 * 268.5e6}, 1, None)
    assert exact["bound"] == "mfma" and exact["peak"] == bench.FP32_PEAK_TFLOPS
    full = json.load(open(os.path.join(os.path.dirname(bench.__file__), "profiles", "r05_bench_enet_full_pool_detail.json")))
    leg = full["secondary"]["c2_bf16x3"]
    assert leg["arithmetic"] == "bf16x3" and "bf16x3" in leg["dtype"] and full["dtype"] == "f32"
    assert leg["score_digest"]["within_tolerance"] and not leg["score_digest"]["bit_identical_to_table"]
    assert leg["score_digest"]["max_abs_diff"] <= 1e-6 and full["score_digest"]["match"] is True
    line = json.loads(bench.compact_line(full, os.path.join(bench.ROOT, "gpurun_out", "bench_detail.json")))
    assert len(json.dumps(line)) <= bench.LINE_LIMIT
    c = line["secondary"]["c2_bf16x3"]
    assert c["dtype"].startswith("f32 via bf16x3") and "opt-in" in c["parity"] and c["roofline_bf16x3"]["bound"] == "hbm"
    assert set(line["secondary"]) == {"c4", "c5", "c2_bf16x3"}
    for k in ("pass_no_overlap_bound", "pass_overlap_bound"):
        assert line["roofline"][k]["images_per_s"] > line["value"]
