"""World-size-2 gloo test (CPU) of the multi-GPU ranking path: strided pool shards, ONE all-gather
of (index, score) pairs, identical top-k on every rank and identical to a single-process run."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_scores(num):
    """deterministic per-example float64 'confidence' with well separated values"""
    rng = np.random.default_rng(123)
    return rng.permutation(num).astype(np.float64) / num + 1e-9 * np.arange(num)


def _worker(rank, world, port, num, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from semanticsegmentationactivelearning_amd import active_learning as al
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scores = _fake_scores(num)
        pos = al.shard_positions(num, rank, world)
        mine = pos[pos >= 0]
        # ragged on purpose: rank 1 "scores" its shard in a different order
        if rank == 1:
            mine = mine[::-1].copy()
        idx = torch.from_numpy(mine)
        sc = torch.from_numpy(scores[mine])
        idx, sc = al._pad_to_common_length(idx, sc, None)
        all_idx, all_sc = al.all_gather_scores(idx, sc)
        unlabelled = np.arange(num)[np.arange(num) % 5 != 0]
        low, uc = al.finish_ranking(all_idx.numpy(), all_sc.numpy(), num, unlabelled, k)
        np.save(os.path.join(out_dir, "low_%d.npy" % rank), np.sort(low))
        np.save(os.path.join(out_dir, "uc_%d.npy" % rank), uc)
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_matches_single_process(tmp_path):
    from semanticsegmentationactivelearning_amd import active_learning as al
    num, k, world = 2975, 128, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num, k, str(tmp_path)), nprocs=world, join=True)
    lows = [np.load(tmp_path / ("low_%d.npy" % r)) for r in range(world)]
    ucs = [np.load(tmp_path / ("uc_%d.npy" % r)) for r in range(world)]
    assert (lows[0] == lows[1]).all() and (ucs[0] == ucs[1]).all()  # identical selection on all ranks
    scores = _fake_scores(num)
    unlabelled = np.arange(num)[np.arange(num) % 5 != 0]
    want_low, want_uc = al.finish_ranking(np.arange(num), scores, num, unlabelled, k)
    assert (np.sort(want_low) == lows[0]).all() and (want_uc == ucs[0]).all()
    # and it is what the reference's tail computes: argpartition of the float32 vector
    conf32 = scores.astype(np.float32)[unlabelled]
    ref = unlabelled[np.argpartition(conf32, k)[:k]]
    assert set(ref.tolist()) == set(lows[0].tolist())


def test_single_process_helpers_are_noops_without_process_group():
    from semanticsegmentationactivelearning_amd import active_learning as al
    idx, sc = torch.arange(5), torch.rand(5, dtype=torch.float64)
    a, b = al._pad_to_common_length(idx, sc, None)
    c, d = al.all_gather_scores(a, b)
    assert torch.equal(c, idx) and torch.equal(d, sc)
