"""Pool scoring and ranking: host-side mirror of the scoring slice of the reference's
``active_learning.py`` (EPSILON :39-40, score ops :229-269, ``rank_confidence`` :682-715,
its caller :776-784).

The per-pixel work (ENet forward, softmax, entropy / margin / confidence, float64 mean) runs in the
HIP kernels behind ``models.ENet.score``; this module keeps the host control flow of the
reference: scatter batch results by example index into a float32 vector, filter the unlabelled
examples, ``np.argpartition`` the ``selection_size`` lowest.  The MI355X addition is pool
sharding: every rank (one process per GPU) scores a strided shard of the pool and ONE RCCL
all-gather of ``(index, score)`` pairs over xGMI per ranking pass rebuilds the full vector on every
rank -- the reference itself is single-process.
"""
import numpy as np

from . import _lib

# Lowest representable (normal) float32 -- reference active_learning.py:39-40
EPSILON = np.finfo(np.float32).tiny

MEASURES = tuple(_lib.MEASURES)  # ("entropy", "margin", "confidence")


class ScoringConfig:
    """The JSON keys the scoring path reads (conf/default_params.json:2,32-35,53-59)."""

    def __init__(self, measure="entropy", selection_size=50, threshold=0.95, batch_size=8,
                 height=None, width=None):
        if measure not in _lib.MEASURES:
            raise NotImplementedError("Uncertainty function not implemented.")
        self.measure = measure
        self.selection_size = int(selection_size)
        self.threshold = float(threshold)
        self.batch_size = int(batch_size)
        self.height = height
        self.width = width

    @classmethod
    def from_params(cls, params):
        al = params["active_learning"]
        net_in = params.get("network", {}).get("input", {})
        return cls(measure=al["measure"], selection_size=al["selection_size"],
                   threshold=al["threshold"], batch_size=params["batch_size"],
                   height=net_in.get("height"), width=net_in.get("width"))


def score_logits(logits, measure="entropy", threshold=0.0, return_label=False, return_mask=False,
                 return_confidence=False):
    """softmax -> {entropy, margin, confidence} -> float64 mean over (H, W) on materialised logits
    (reference :239-263): ``pseudo_mean_confidence`` [N] float64, plus optionally ``pseudo_label``
    (uint8 argmax, :234-236), ``pseudo_mask`` (conf < threshold -> 0 else 1, :265-269) and the
    per-pixel ``pseudo_confidence``.  Raises NotImplementedError for an unknown measure (:259-260)."""
    if measure not in _lib.MEASURES:
        raise NotImplementedError("Uncertainty function not implemented.")
    torch = _lib.require_gpu()
    x = _lib.as_device_f32(logits)
    if x.dim() != 4:
        raise ValueError("logits must be [N,H,W,classes]")
    n, h, w, k = x.shape
    L = _lib.lib()
    with torch.cuda.device(x.device):
        nbytes = L.ssal_score_workspace_bytes(n, h, w)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
        scores = torch.empty((n,), dtype=torch.float64, device=x.device)
        label = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if return_label else None
        mask = torch.empty((n, h, w), dtype=torch.uint8, device=x.device) if return_mask else None
        conf = torch.empty((n, h, w), dtype=torch.float32, device=x.device) if return_confidence else None
        _lib.check(L.ssal_score_logits_nhwc(
            _lib.dev_ptr(x), n, h, w, k, _lib.MEASURES[measure], float(threshold),
            _lib.dev_ptr(scores), _lib.dev_ptr(label), _lib.dev_ptr(mask), _lib.dev_ptr(conf),
            _lib.dev_ptr(ws), ws.numel(), _lib.stream_ptr()))
    if return_label or return_mask or return_confidence:
        return scores, {"label": label, "mask": mask, "confidence": conf}
    return scores


def select_lowest(unlabelled_confidence, selection_size):
    """``np.argpartition(conf, k)[:k]`` -- the k lowest-confidence positions as an unordered set
    (reference :707-712).  The reference raises when ``selection_size == len(unlabelled)`` (kth out
    of bounds, a known defect); here that case returns every position."""
    conf = np.asarray(unlabelled_confidence)
    k = int(np.minimum(len(conf), selection_size))
    if k <= 0:
        return np.zeros((0,), dtype=np.int64)
    if k >= len(conf):
        return np.arange(len(conf), dtype=np.int64)
    return np.argpartition(conf, k)[:k].astype(np.int64)


def shard_positions(num_examples, rank, world_size):
    """Strided shard of pool positions for one rank; every rank gets the same count after padding
    with -1 sentinels (2975 = 8*371 + 7 -> 372 per rank)."""
    per = (num_examples + world_size - 1) // world_size
    pos = np.arange(rank, num_examples, world_size, dtype=np.int64)
    pad = np.full((per - len(pos),), -1, dtype=np.int64)
    return np.concatenate([pos, pad])


def all_gather_scores(local_index, local_score, group=None):
    """ONE all-gather (RCCL over xGMI for GPU tensors, gloo for CPU tensors) of each rank's
    ``(index int64, score float64)`` shard; returns the concatenation (sentinel index -1 kept)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_index, local_score
    return _gather_pairs(local_index, local_score, group)


def _gather_pairs(local_index, local_score, group=None):
    """the collective itself (also what the single-rank RCCL smoke test drives): pack the index next to the score in a
    float64 pair -> [per, 2], one ``all_gather_into_tensor``"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    packed = torch.stack([local_index.to(torch.float64), local_score.to(torch.float64)], dim=1).contiguous()
    home = packed.device
    if packed.is_cuda and dist.get_backend(group) == "gloo":
        packed = packed.cpu()  # rehearsals of the multi-rank path on one GPU run over gloo: collectives on CPU tensors
    gathered = torch.empty((world * packed.shape[0], 2), dtype=torch.float64, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    gathered = gathered.to(home)
    return gathered[:, 0].to(torch.int64), gathered[:, 1]


def prefetch_to_device(batches, depth=2):
    """Host batches -> device batches, copied ``depth`` batches ahead on a side HIP stream, so that the
    host-to-device copy of batch i+1 overlaps the kernels of batch i (the reference gets
    the same effect from ``tf.data`` prefetching, ``tensortools/input.py:195``).  ``batches`` yields
    ``(images, example_indices)`` with images as numpy / CPU-torch arrays (uint8 frames or float32) or tensors
    already on the GPU (passed through)."""
    import collections
    torch = _lib.require_gpu()
    copy_stream = torch.cuda.Stream()
    queue = collections.deque()

    def stage(item):
        images, indices = item
        if isinstance(images, torch.Tensor) and images.is_cuda:
            return images, indices, None, None
        host = images if isinstance(images, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(images))
        host = host.contiguous()
        # The copy runs on the side stream, i.e. NOT behind the kernels of the batches already enqueued on the
        # compute stream.  Pinned sources are copied asynchronously; pageable ones block the host for the copy
        # (the runtime stages them through its own pinned buffers) while the GPU keeps computing.  Pinning
        # per batch here would cost more than it saves (page-locking 50-200 MB takes milliseconds).
        with torch.cuda.stream(copy_stream):
            dev = host.cuda(non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        # page-locked ring slot of a tensortools.input.InputStage (found by the address of the batch's memory, so
        # sliced / re-wrapped batches resolve too): the stage does not rewrite it before `done`
        from .tensortools import input as _input
        _input.copy_issued(host, done)
        return dev, indices, done, host  # the host buffer must outlive the copy

    def release(entry):
        dev, indices, done, _pinned = entry
        if done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            dev.record_stream(cur)  # allocated on the copy stream, consumed on this one
        return dev, indices

    for item in batches:
        queue.append(stage(item))
        if len(queue) > depth:
            yield release(queue.popleft())
    while queue:
        yield release(queue.popleft())


def pad_to_length(index, score, length):
    """Local (collective-free) padding of one rank's ``(index, score)`` shard to ``length`` entries with the
    ``(-1, +inf)`` sentinel.  ``shard_positions`` gives every rank ``ceil(num_examples / world)`` positions, so
    a rank that scored its whole shard pads to exactly that length without asking the others."""
    import torch
    pad = int(length) - index.numel()
    if pad < 0:
        raise ValueError("shard has %d entries, more than the per-rank length %d" % (index.numel(), length))
    if pad > 0:
        index = torch.cat([index, torch.full((pad,), -1, dtype=torch.int64, device=index.device)])
        score = torch.cat([score, torch.full((pad,), float("inf"), dtype=torch.float64, device=score.device)])
    return index, score


def rank_confidence(net, batches, num_examples, unlabelled, selection_size, measure="entropy",
                    group=None, prefetch=0, ragged=False, arithmetic="f32"):
    """Mirror of ``rank_confidence()`` (reference :682-715).

    ``batches`` yields ``(images NHWC float32 or uint8, example_indices)``; on a multi-GPU job each rank
    passes only its own shard (``shard_positions``).  ``prefetch`` > 0 copies host batches that many batches
    ahead on a side stream (``prefetch_to_device``).  Returns ``(low_conf_examples, unlabelled_confidence)``:
    the ids (into the full example list) of the ``selection_size`` least confident unlabelled examples and
    the float32 confidence of every unlabelled example (the reference feeds it to a histogram summary,
    :781-784).

    ``arithmetic`` is handed to ``net.score`` ("f32": exact fp32, the default; "bf16x3": ENet's opt-in split-operand mode).

    Collectives per ranking pass: exactly ONE all-gather of ``(index, score)`` pairs.  Shards handed out by
    ``shard_positions`` hold at most ``ceil(num_examples / world)`` examples, so each rank pads locally to that
    length (+ one flag entry).  A rank whose shard is longer raises ``ValueError`` -- on EVERY rank, after the
    collective, so nobody is left blocking in it.  ``ragged=True`` is for callers that split the pool some other way
    (shard lengths unknown to the other ranks): it costs one extra all-reduce(MAX) to agree on the length."""
    torch = _lib.require_gpu()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _lib.warn_if_few_hw_queues()  # caller + 2 chains + prefetch copy + RCCL = 5 streams (once per process)
    idx_chunks, score_chunks = [], []
    if prefetch > 0:
        batches = prefetch_to_device(batches, depth=prefetch)
    for images, indices in batches:
        # [n] float64 on device, stream-ordered; arithmetic: "f32" (the reference's, default) or ENet's opt-in "bf16x3"
        s = net.score(images, measure=measure) if arithmetic == "f32" else net.score(images, measure=measure, arithmetic=arithmetic)
        score_chunks.append(s)
        idx_chunks.append(torch.as_tensor(np.asarray(indices, dtype=np.int64), device=s.device))
    if score_chunks:
        local_score = torch.cat(score_chunks)
        local_index = torch.cat(idx_chunks)
    else:
        dev = torch.device("cuda", torch.cuda.current_device())
        local_score = torch.zeros((0,), dtype=torch.float64, device=dev)
        local_index = torch.zeros((0,), dtype=torch.int64, device=dev)
    return merge_and_rank(local_index, local_score, num_examples, unlabelled, selection_size, group, ragged)


def merge_and_rank(local_index, local_score, num_examples, unlabelled, selection_size, group=None, ragged=False):
    """the collective + host tail of a ranking pass (shared by ``rank_confidence`` and ``bench.py``)"""
    import torch.distributed as dist
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    overflow = False
    if world > 1:
        if ragged:
            local_index, local_score = _pad_to_common_length(local_index, local_score, group)
        else:
            # precondition: len(shard) <= ceil(num_examples / world) (what shard_positions hands out).  A rank that
            # violates it must not raise BEFORE the collective (the others would block in the all-gather until the
            # backend times out): it contributes an all-sentinel shard of the agreed length whose extra last entry
            # carries the flag (index -1, score -inf; +inf = fine), and EVERY rank raises after the collective.
            per = (num_examples + world - 1) // world
            overflow = local_index.numel() > per
            if overflow:
                local_index, local_score = local_index[:0], local_score[:0]
            local_index, local_score = pad_to_length(local_index, local_score, per + 1)
            if overflow:
                local_score = local_score.clone()
                local_score[-1] = float("-inf")
    all_index, all_score = all_gather_scores(local_index, local_score, group)
    all_index, all_score = all_index.cpu().numpy(), all_score.cpu().numpy()
    if world > 1 and not ragged:
        bad = np.nonzero((all_index < 0) & np.isneginf(all_score))[0]
        if len(bad):
            raise ValueError("rank(s) %s handed merge_and_rank a shard longer than ceil(num_examples / world) = %d "
                             "entries; split the pool with shard_positions() or pass ragged=True"
                             % (sorted(set((bad // (per + 1)).tolist())), per))
    return finish_ranking(all_index, all_score, num_examples, unlabelled, selection_size)


def _pad_to_common_length(index, score, group):
    """Ragged callers only: all_gather_into_tensor needs equal shard lengths, and nobody knows the longest
    one, so agree on it with one all-reduce(MAX), then pad with the (-1, +inf) sentinel."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return index, score
    on_cpu = index.is_cuda and dist.get_backend(group) == "gloo"
    n = torch.tensor([index.numel()], dtype=torch.int64, device="cpu" if on_cpu else index.device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX, group=group)
    return pad_to_length(index, score, int(n.item()))


def finish_ranking(all_index, all_score, num_examples, unlabelled, selection_size):
    """Host tail of rank_confidence (reference :685,700,705-715): scatter into a float32 vector by
    example index (the float64 -> float32 rounding happens here, like ``confidence[batch_indices] =
    batch_confidence``), filter the unlabelled subset, pick the lowest ``selection_size``."""
    confidence = np.zeros(num_examples, dtype=np.float32)
    valid = all_index >= 0
    confidence[all_index[valid]] = all_score[valid]
    unlabelled = np.asarray(unlabelled, dtype=np.int64)
    unlabelled_confidence = confidence[unlabelled]
    example_indices = select_lowest(unlabelled_confidence, selection_size)
    low_conf_examples = unlabelled[example_indices]
    return low_conf_examples, unlabelled_confidence
