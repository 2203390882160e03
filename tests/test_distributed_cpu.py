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


class _CollectiveCounter:
    """counts every torch.distributed collective issued while active"""
    NAMES = ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "reduce", "all_to_all",
             "gather", "scatter", "reduce_scatter", "barrier", "all_gather_object")

    def __enter__(self):
        self.calls, self._orig = [], {}
        for nm in self.NAMES:
            if hasattr(dist, nm):
                self._orig[nm] = getattr(dist, nm)
                setattr(dist, nm, self._wrap(nm, self._orig[nm]))
        return self

    def _wrap(self, nm, fn):
        def inner(*a, **kw):
            self.calls.append(nm)
            return fn(*a, **kw)
        return inner

    def __exit__(self, *exc):
        for nm, fn in self._orig.items():
            setattr(dist, nm, fn)


def _worker(rank, world, port, num, k, out_dir, ragged):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from semanticsegmentationactivelearning_amd import active_learning as al
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scores = _fake_scores(num)
        pos = al.shard_positions(num, rank, world)
        mine = pos[pos >= 0]
        # rank 1 "scores" its shard in a different order; in the ragged case it also owns fewer examples than
        # shard_positions would give it (rank 0 takes them over), which the other rank cannot know
        if rank == 1:
            mine = mine[::-1].copy()
        if ragged:
            extra = al.shard_positions(num, 1, world)
            extra = extra[extra >= 0][:100]
            mine = np.concatenate([mine, extra]) if rank == 0 else mine[:-100]
        idx = torch.from_numpy(mine)
        sc = torch.from_numpy(scores[mine])
        unlabelled = np.arange(num)[np.arange(num) % 5 != 0]
        with _CollectiveCounter() as cc:
            low, uc = al.merge_and_rank(idx, sc, num, unlabelled, k, ragged=ragged)
        want = ["all_reduce", "all_gather_into_tensor"] if ragged else ["all_gather_into_tensor"]
        assert cc.calls == want, cc.calls
        np.save(os.path.join(out_dir, "low_%d.npy" % rank), np.sort(low))
        np.save(os.path.join(out_dir, "uc_%d.npy" % rank), uc)
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("ragged", [False, True])
def test_two_rank_gather_matches_single_process(tmp_path, ragged):
    """shard_positions shards: exactly ONE collective (the all-gather) per ranking pass; ragged shards pay one
    extra all-reduce(MAX) to agree on the length.  Same selection on every rank and as a single process."""
    from semanticsegmentationactivelearning_amd import active_learning as al
    num, k, world = 2975, 128, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, num, k, str(tmp_path), ragged), nprocs=world, join=True)
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
    e, f = al.pad_to_length(idx, sc, 8)
    assert e.tolist() == [0, 1, 2, 3, 4, -1, -1, -1] and torch.isinf(f[5:]).all() and torch.equal(f[:5], sc)
    with pytest.raises(ValueError):
        al.pad_to_length(idx, sc, 4)


def _overflow_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    from semanticsegmentationactivelearning_amd import active_learning as al
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        num = 21  # ceil(21 / 2) = 11 per rank; rank 1 shows up with 13 entries
        mine = np.arange(rank, num, world)
        if rank == 1:
            mine = np.concatenate([mine, [0, 2, 4]])
        idx, sc = torch.from_numpy(mine), torch.from_numpy(_fake_scores(num)[mine])
        with _CollectiveCounter() as cc:
            try:
                al.merge_and_rank(idx, sc, num, np.arange(num), 5)
                verdict = "no error"
            except ValueError as e:
                verdict = "ValueError: %s" % e
        open(os.path.join(out_dir, "verdict_%d.txt" % rank), "w").write("%s|%s" % (cc.calls, verdict))
    finally:
        dist.destroy_process_group()


def test_shard_longer_than_agreed_length_fails_on_every_rank_after_the_collective(tmp_path):
    """a rank whose shard exceeds ceil(num_examples / world) must not raise before the all-gather (the other
    rank would hang in it): both ranks run the ONE collective, then both raise ValueError naming the offending rank"""
    world = 2
    mp.spawn(_overflow_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        calls, verdict = open(tmp_path / ("verdict_%d.txt" % r)).read().split("|", 1)
        assert calls == "['all_gather_into_tensor']", calls
        assert verdict.startswith("ValueError") and "[1]" in verdict, verdict


def _c3_worker(rank, world, port, num, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    from semanticsegmentationactivelearning_amd import active_learning as al
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        scores = _fake_scores(num)
        pos = al.shard_positions(num, rank, world)
        assert len(pos) == 372 and int((pos < 0).sum()) == (0 if rank < num % world else 1)  # 2975 = 8 * 371 + 7
        mine = pos[pos >= 0]
        # the bench's strong-scaling pass: every rank runs the SAME 47 steps of 8 (a short shard's last batch wraps onto
        # batch 0: duplicates carry the same bits); then pads locally to 47 * 8 = 376 entries and all-gathers once
        per_rank = (num + world - 1) // world
        steps = (per_rank + 7) // 8
        assert (per_rank, steps) == (372, 47)
        batches = [mine[b * 8:(b + 1) * 8] for b in range((len(mine) + 7) // 8)]
        order = np.concatenate([batches[s % len(batches)] for s in range(steps)])
        idx, sc = al.pad_to_length(torch.from_numpy(order), torch.from_numpy(scores[order]), steps * 8)
        with _CollectiveCounter() as cc:
            all_idx, all_sc = al.all_gather_scores(idx, sc)
        assert cc.calls == ["all_gather_into_tensor"], cc.calls
        assert tuple(all_idx.shape) == (world * steps * 8,)
        low, uc = al.finish_ranking(all_idx.numpy(), all_sc.numpy(), num, np.arange(num), k)
        # and the library's own one-call form on the un-wrapped shard: exactly one collective as well
        with _CollectiveCounter() as cc2:
            low2, uc2 = al.merge_and_rank(torch.from_numpy(mine), torch.from_numpy(scores[mine]), num, np.arange(num), k)
        assert cc2.calls == ["all_gather_into_tensor"], cc2.calls
        assert set(low.tolist()) == set(low2.tolist()) and (uc == uc2).all()
        np.save(os.path.join(out_dir, "low_%d.npy" % rank), np.sort(low))
        np.save(os.path.join(out_dir, "uc_%d.npy" % rank), uc)
    finally:
        dist.destroy_process_group()


def test_world_8_exact_c3_split_one_collective_top_128(tmp_path):
    """BASELINE configs[2] rehearsed on CPU (gloo, world_size = 8): 2975 examples -> 372 per rank (ranks 0..6 hold 372
    examples, rank 7 holds 371 + one sentinel), k = 128, `strong_steps` = 47 on every rank, ONE all-gather, identical
    selection on all 8 ranks and equal to the single-process / reference argpartition result"""
    import bench
    from semanticsegmentationactivelearning_amd import active_learning as al
    num, k, world = bench.POOL, bench.TOP_K, 8
    assert (num, k) == (2975, 128)
    port = _free_port()
    mp.spawn(_c3_worker, args=(world, port, num, k, str(tmp_path)), nprocs=world, join=True)
    lows = [np.load(tmp_path / ("low_%d.npy" % r)) for r in range(world)]
    ucs = [np.load(tmp_path / ("uc_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert (lows[r] == lows[0]).all() and (ucs[r] == ucs[0]).all()
    scores = _fake_scores(num)
    ref = np.argpartition(scores.astype(np.float32), k)[:k]
    assert set(ref.tolist()) == set(lows[0].tolist()) and len(lows[0]) == k
    want_low, want_uc = al.finish_ranking(np.arange(num), scores, num, np.arange(num), k)
    assert (np.sort(want_low) == lows[0]).all() and (want_uc == ucs[0]).all()
