"""The sharded ranking pass with REAL scores: two ranks (one process each, sharing the one GPU of the box,
collectives over gloo) score their `shard_positions` shard of a small synthetic pool with `ENet.score` through
`rank_confidence`, and every rank must select exactly the examples the single-process oracle selects.  What this
cannot exercise is RCCL itself (one GPU): the collective here runs on CPU tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

NUM, K, H, W, BS = 22, 6, 32, 64, 4  # 22 examples over 2 ranks: 11 each, last batch short


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from helpers import make_model
    from semanticsegmentationactivelearning_amd import active_learning as al, synthetic as syn
    from test_distributed_cpu import _CollectiveCounter
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        net, _ = make_model(19, 3, seed=0)
        pos = al.shard_positions(NUM, rank, world)
        mine = pos[pos >= 0]
        batches = [(syn.synth_frames_f32(mine[i:i + BS] + 100, H, W, 3), mine[i:i + BS]) for i in range(0, len(mine), BS)]
        unlabelled = np.arange(NUM)[np.arange(NUM) % 4 != 1]
        with _CollectiveCounter() as cc:
            low, uconf = al.rank_confidence(net, batches, NUM, unlabelled, K, measure="entropy")
        assert cc.calls == ["all_gather_into_tensor"], cc.calls  # ONE collective per ranking pass
        np.save(os.path.join(out_dir, "low_%d.npy" % rank), np.sort(low))
        np.save(os.path.join(out_dir, "uc_%d.npy" % rank), uconf)
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_real_scores_match_oracle(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import frames, make_model
    from oracle import enet_oracle as orc
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    lows = [np.load(tmp_path / ("low_%d.npy" % r)) for r in range(world)]
    ucs = [np.load(tmp_path / ("uc_%d.npy" % r)) for r in range(world)]
    assert (lows[0] == lows[1]).all() and (ucs[0] == ucs[1]).all()  # identical selection on every rank
    _, P = make_model(19, 3, seed=0)
    want_mean = orc.score_images(P, frames(np.arange(NUM) + 100, H, W, 3), "entropy")[0]
    unlabelled = np.arange(NUM)[np.arange(NUM) % 4 != 1]
    want_low, want_u = orc.rank_lowest(want_mean, unlabelled, K)
    srt = np.sort(want_u)
    assert srt[K] - srt[K - 1] > 1e-5, "fixture must separate the k-th boundary"
    assert sorted(want_low.tolist()) == lows[0].tolist()
    assert np.abs(ucs[0] - want_u).max() <= 1e-6


def _rccl_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from semanticsegmentationactivelearning_amd import active_learning as al
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm (bench.py's call)
    try:
        idx = torch.arange(11, device=dev)
        sc = torch.linspace(0.1, 0.9, 11, dtype=torch.float64, device=dev)
        gi, gs = al._gather_pairs(idx, sc)  # all_gather_into_tensor on DEVICE tensors through RCCL
        t = torch.tensor([3.5, 11.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the reductions bench.py issues after the timed region
        dist.barrier()
        torch.cuda.synchronize()
        ok = torch.equal(gi, idx) and torch.equal(gs, sc) and t.tolist() == [3.5, 11.0] and gi.is_cuda
        open(os.path.join(out_dir, "rccl_ok"), "w").write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def test_rccl_backend_executes_with_one_rank(tmp_path):
    """One GPU cannot host two RCCL ranks, so this is NOT a multi-GPU test; it proves that the `nccl` branch
    (init_process_group with device_id, all_gather_into_tensor / all_reduce / barrier on device tensors) loads RCCL and
    executes on this machine -- the part the gloo rehearsals cannot reach."""
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "rccl_ok").read() == "1"


def _run_bench(extra, nproc, tmp_path, tag):
    import json
    import subprocess
    env = dict(os.environ, SSAL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--height", "64", "--width", "128", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-secondary"]
    bench = os.path.join(ROOT, "bench.py")
    if nproc == 1:
        cmd = [sys.executable, bench, "--gpus", "1"] + common + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), bench, "--gpus", str(nproc)] + common + extra
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]  # stdout carries exactly ONE JSON line
    return json.loads(lines[0])


def test_bench_two_ranks_strong_scaling_end_to_end(tmp_path):
    """`bench.py` launched exactly as the driver does it for N = 2 (torch.distributed.run, one process per
    rank; here both ranks share the one GPU and the collectives run over gloo) defaults to STRONG scaling: the whole
    2975-frame pool split over the ranks, the weak figure beside it, and the same top-128 selection / the same score digest
    as a single-process pass over the pool (small frames, so the pass takes seconds)."""
    two = _run_bench(["--steps", "5"], 2, tmp_path, "n2")
    one = _run_bench(["--steps", "372"], 1, tmp_path, "n1")
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["config"]["frames_scored"] == 2975
    assert two["steps"] == 186 and two["requested_steps"] == 5  # ceil(ceil(2975 / 2) / 8)
    assert two["weak"]["scaling"] == "weak" and two["weak"]["steps"] == 5 and two["weak"]["frames_scored"] == 2 * 5 * 8
    assert one["scaling"] == "weak" and one["steps"] == 372 and one["config"]["frames_scored"] == 2975
    assert two["top_k_checksum"] is not None and two["top_k_checksum"] == one["top_k_checksum"]
    assert two["score_digest"]["frames"] == 2975 and two["score_digest"]["sha256"] == one["score_digest"]["sha256"]
    assert two["score_digest"]["match"] is None  # no committed table for the small rehearsal frames


def test_bench_missing_rank_exits_nonzero_instead_of_hanging(tmp_path):
    """a rank that never arrives: the rendezvous times out and the process exits non-zero"""
    import subprocess
    env = dict(os.environ, SSAL_DIST_BACKEND="gloo", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--height", "64", "--width", "128",
                        "--dist-timeout", "8", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, timeout=120)
    assert r.returncode != 0
    assert b"{\"metric\"" not in r.stdout


def _prefetch_worker(rank, world, port, out_dir, pool_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import glob
    import torch.distributed as dist
    from helpers import make_model
    from semanticsegmentationactivelearning_amd import _lib, active_learning as al
    from semanticsegmentationactivelearning_amd.tensortools import InputStage
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert _lib.hw_queues_ok(), os.environ.get("GPU_MAX_HW_QUEUES")  # the package default (8), applied at import
        net, _ = make_model(19, 3, seed=0)
        files = np.array(sorted(glob.glob(os.path.join(pool_dir, "*.tfrecord"))))
        pos = al.shard_positions(len(files), rank, world)
        mine = pos[pos >= 0]
        stage = InputStage(input_shape=[64, 64], image_dtype=np.uint8, pin_memory=True, pin_buffers=4)
        stage.add_dataset_from_placeholders("train", files[mine], np.zeros(len(mine), dtype=bool), mine, batch_size=3)
        stage.init_iterator("train", None, None)

        def batches():
            for image, label, mask, labelled, index in stage:
                yield image, index
        low, uconf = al.rank_confidence(net, batches(), len(files), np.arange(len(files)), 4, prefetch=2)
        np.save(os.path.join(out_dir, "plow_%d.npy" % rank), np.sort(low))
        np.save(os.path.join(out_dir, "puc_%d.npy" % rank), uconf)
    finally:
        dist.destroy_process_group()


def test_two_ranks_prefetch_from_tfrecords_five_streams(tmp_path):
    """the five-stream case -- each of two ranks (sharing the one GPU, collectives over gloo) runs
    rank_confidence(prefetch=2) on ITS shard of a pool of real TFRecords: caller's stream + 2 image-group chains + the
    prefetch copy stream (+ the collective).  Both ranks must select what a single-process float32 pass selects."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import frames, make_model
    from oracle import enet_oracle as orc
    from test_input_cpu import write_pool
    num, world = 13, 2
    pool = tmp_path / "pool"
    pool.mkdir()
    write_pool(str(pool), num, 64, 64, with_label=False)
    mp.spawn(_prefetch_worker, args=(world, _free_port(), str(tmp_path), str(pool)), nprocs=world, join=True)
    lows = [np.load(tmp_path / ("plow_%d.npy" % r)) for r in range(world)]
    ucs = [np.load(tmp_path / ("puc_%d.npy" % r)) for r in range(world)]
    assert (lows[0] == lows[1]).all() and (ucs[0] == ucs[1]).all()
    _, P = make_model(19, 3, seed=0)
    want = orc.score_images(P, frames(np.arange(num), 64, 64, 3), "entropy")[0]
    want_low, want_u = orc.rank_lowest(want, np.arange(num), 4)
    srt = np.sort(want_u)
    assert srt[4] - srt[3] > 1e-5, "fixture must separate the k-th boundary"
    assert sorted(want_low.tolist()) == lows[0].tolist()
    assert np.abs(ucs[0] - want_u).max() <= 1e-6
