#!/usr/bin/env python3
"""bench.py -- unlabelled-pool images/sec scored (ENet, 1024x2048) on N MI355X GPUs.

Workload (BASELINE.json configs[1] / [2]): a pool of 2975 synthetic Cityscapes-shaped frames
(1024x2048x3 fp32 NHWC, device-resident before the timed region), ENet(19 classes) with seeded
synthetic weights, entropy acquisition, batches of 8 (reference conf/*.json:2).  One "step" = one
batch of 8 frames through the fused path: ENet forward + per-pixel softmax-entropy + float64
per-image mean.  With N > 1 the pool is sharded over the ranks (one process per GPU, weak scaling:
every rank scores `steps` batches of its own shard) and the per-image (index, score) pairs are
merged by ONE RCCL all-gather followed by the float32 scatter + top-128 argpartition on every
rank; that merge is inside the timed region.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant kernel,
HIP-event timed live on the launch stream in a separate profiling pass) and `cpu_baseline` (the
torch-CPU restatement of the reference path, timed on this box's host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of the GPU pool only supports dmabuf IPC: RCCL / device-tensor sharing across the ranks of one node
# needs this before the HIP runtime comes up (it is exported on the pool already; kept here for any other launcher)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

POOL = 2975          # Cityscapes train split size (BASELINE.json configs[1])
TOP_K = 128          # BASELINE.json configs[2]
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak


def log(msg):
    """progress on stderr (stdout carries exactly one JSON line)"""
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=372)   # 372 batches of 8 = the whole 2975-frame pool
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--classes", type=int, default=19)
    ap.add_argument("--channels", type=int, default=3)
    ap.add_argument("--model", choices=["enet", "icnet"], default="enet",
                    help="enet: BASELINE configs[1] (the metric's workload); icnet: configs[3] (ICNet multi-scale, use "
                         "--measure margin), architecture pinned in ICNET_SPEC.md")
    ap.add_argument("--measure", default=None, help="entropy | margin | confidence (default: entropy, margin for icnet)")
    ap.add_argument("--input-dtype", choices=["f32", "u8"], default="f32",
                    help="resident frames: float32 in [0,1] (the reference's model input) or the decoded uint8 "
                         "frames, converted inside the Initial kernel (same bits out)")
    ap.add_argument("--resident-gib", type=float, default=96.0,
                    help="cap on device memory used for resident input frames (wraps beyond it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank scores --steps batches of its own shard (per-GPU work fixed); strong: the "
                         "whole 2975-frame pool is split over the ranks (total work fixed, --steps is ignored and "
                         "reported as the batches rank 0 ran), value = 2975 / t")
    ap.add_argument("--allow-nondefault-knobs", action="store_true",
                    help="measurement runs only (tools/*.sh with a -DSSAL_MEASURE library): time the library although "
                         "ssal_debug_get_knobs() says a switch is off its default; the JSON line still reports them")
    return ap.parse_args()


def cpu_baseline(P, h, w, c, measure, budget_s, c1=None, model="enet"):
    """torch-CPU restatement of the reference path (oracle/torch_restatement.py) on a bounded
    sample: single 1024x2048 frames, 1 warm-up + as many repeats as fit the budget (>= 2)."""
    import torch
    from oracle import torch_restatement as tr
    from semanticsegmentationactivelearning_amd import synthetic as syn
    from semanticsegmentationactivelearning_amd._lib import usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    log("cpu baseline: %d threads (affinity %d, cpu_count %s)" % (cores, len(os.sched_getaffinity(0)), os.cpu_count()))
    x = syn.synth_frames_f32([0], h, w, c)
    score_images = tr.icnet_score_images if model == "icnet" else tr.score_images
    t0 = time.perf_counter()
    score_images(P, x, measure)  # warm-up
    log("cpu baseline warm-up %.2f s" % (time.perf_counter() - t0))
    times = []
    t_all = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        score_images(P, x, measure)
        times.append(time.perf_counter() - t0)
        log("cpu baseline run %d: %.2f s" % (len(times), times[-1]))
        if len(times) >= 20:
            break
    med = float(np.median(times))
    out = {"value": 1.0 / med, "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
           "sample": "%d x 1 frame %dx%dx%d forward+%s score, torch-CPU fp32 restatement of the reference "
                     "TF path (TensorFlow itself is not installable here), median of %d runs after 1 warm-up"
                     % (len(times), h, w, c, measure, len(times))}
    if c1 is not None:
        # BASELINE.json configs[0] / SURVEY 8(d): the reference's own CPU-runnable case, timed exactly:
        # 4 x 256x512x3 frames -> forward + entropy score + top-1 select
        P1, k1 = c1
        x1 = syn.synth_frames_f32([0, 1, 2, 3], 256, 512, 3)
        tr.score_images(P1, x1, "entropy")
        t1 = []
        while len(t1) < 3 or sum(t1) < 3.0:
            t0 = time.perf_counter()
            mean = tr.score_images(P1, x1, "entropy")[0]
            np.argpartition(np.asarray(mean, dtype=np.float32), 1)[:1]
            t1.append(time.perf_counter() - t0)
            if len(t1) >= 30:
                break
        m1 = float(np.median(t1))
        log("cpu baseline C1 (4 x 256x512): %.3f s" % m1)
        out["c1"] = {"value": 4.0 / m1, "unit": "images/s", "cores": int(torch.get_num_threads()), "kind": "port",
                     "sample": "configs[0]: %d x (4 frames 256x512x3, K=%d: forward + entropy + top-1), median"
                               % (len(t1), k1)}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import semanticsegmentationactivelearning_amd as ssal
    from semanticsegmentationactivelearning_amd import _lib, active_learning as al, synthetic as syn

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # SSAL_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks share
    # the cards round-robin, collectives run on CPU tensors); the real run is one rank per GPU over RCCL
    backend = os.environ.get("SSAL_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # launched through torch.distributed.run: initialise the process group even for ONE rank, so that the single-GPU
    # rehearsal of the launcher command exercises the RCCL branch (init with device_id, barriers, reductions)
    use_dist = world > 1 or os.environ.get("TORCHELASTIC_RUN_ID") is not None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints its banner (host name, library path) on the C-level stdout when the communicator comes up:
        # stdout carries exactly ONE JSON line, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "gloo":
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    knobs = _lib.get_knobs()
    if not knobs["defaults"] and not args.allow_nondefault_knobs:
        raise SystemExit("refusing to time a library whose switches are not at their shipping values: %s" % knobs)

    h, w, c, bs = args.height, args.width, args.channels, args.batch
    if args.measure is None:
        args.measure = "margin" if args.model == "icnet" else "entropy"
    if args.model == "icnet":
        net = ssal.ICNet(args.classes)
        net.build((None, None, None, c))
        syn.randomize_icnet(net, seed=0)
    else:
        net = ssal.ENet(args.classes)
        net.build((None, None, None, c))
        syn.randomize_enet(net, seed=0)
    model_name = "ICNet" if args.model == "icnet" else "ENet"

    # ---- this rank's shard of the pool, device-resident before the clock starts -------------------
    positions = al.shard_positions(POOL, rank, world)
    positions = positions[positions >= 0]
    n_batches_shard = (len(positions) + bs - 1) // bs
    in_dtype = torch.uint8 if args.input_dtype == "u8" else torch.float32
    bytes_per_batch = bs * h * w * c * (1 if args.input_dtype == "u8" else 4)
    max_resident = max(1, int(args.resident_gib * 2 ** 30 // bytes_per_batch))
    if args.scaling == "strong":
        # total work fixed: this rank scores its whole shard of the 2975-frame pool, once
        args.steps = n_batches_shard
        args.warmup = min(args.warmup, 2)
    need = min(args.steps + args.warmup, n_batches_shard)
    n_resident = min(need, max_resident)
    batches = []
    for b in range(n_resident):
        ids = positions[b * bs:(b + 1) * bs]
        buf = torch.empty((len(ids), h, w, c), dtype=in_dtype, device=dev)
        # strided shard: frame ids are not consecutive when world > 1 -> one generator call per frame
        if world == 1:
            syn.synth_frames_device(int(ids[0]), len(ids), h, w, c, out=buf)
        else:
            for j, f in enumerate(ids):
                syn.synth_frames_device(int(f), 1, h, w, c, out=buf[j:j + 1])
        batches.append((buf, torch.as_tensor(ids, device=dev)))
    torch.cuda.synchronize()
    log("%d resident batches of %d frames (%.1f GiB) generated on device" %
        (n_resident, bs, n_resident * bytes_per_batch / 2 ** 30))

    def run_steps(k, first):
        idx_chunks, score_chunks, frames = [], [], 0
        for s in range(k):
            xb, ib = batches[(first + s) % n_resident]
            score_chunks.append(net.score(xb, measure=args.measure))
            idx_chunks.append(ib)
            frames += xb.shape[0]
        return torch.cat(idx_chunks), torch.cat(score_chunks), frames

    def merge_and_select(index, score):
        # ONE collective: every rank ran the same number of steps, so each pads its (index, score) shard locally
        # to steps * batch entries (a shard's last batch may be short) and all-gathers it
        if use_dist:
            index, score = al.pad_to_length(index, score, max(args.steps, args.warmup, 1) * bs)
        all_index, all_score = al.all_gather_scores(index, score)
        return al.finish_ranking(all_index.cpu().numpy(), all_score.cpu().numpy(), POOL,
                                 np.arange(POOL), TOP_K)

    # ---- warm-up (untimed) ------------------------------------------------------------------------
    i0, s0, _ = run_steps(max(args.warmup, 1), 0)
    merge_and_select(i0, s0)
    torch.cuda.synchronize()
    log("warm-up done")

    # ---- timed region: exactly --steps steps, barrier + synchronize on both sides -----------------
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    index, score, frames = run_steps(args.steps, 0 if args.scaling == "strong" else args.warmup)
    low, _ = merge_and_select(index, score)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    log("timed region: %d steps in %.3f s on this rank" % (args.steps, elapsed))
    t = torch.tensor([elapsed, float(frames)], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
    if use_dist:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        total_frames = float(t[1])
    else:
        total_frames = float(frames)

    result = None
    if rank == 0:
        value = total_frames / elapsed
        result = {
            "metric": "unlabelled-pool images/sec scored (%s, %dx%d)" % (model_name, h, w),
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s pool of %d synthetic %dx%dx%d frames, %s acquisition, "
                                   "batch %d, K=%d, top-%d select%s" % ("configs[3]: ICNet multi-scale (1/4, 1/2, 1; ICNET_SPEC.md)"
                                                                        if args.model == "icnet" else "configs[1]: ENet",
                                                                        POOL, h, w, c, args.measure, bs, args.classes, TOP_K,
                                                                        ", uint8 resident frames" if args.input_dtype == "u8" else ""),
                       "frames_scored": int(total_frames), "resident_batches_per_rank": n_resident,
                       "sharding": "strided pool shard per rank, one all-gather of (index, score)"},
            "knobs": knobs,
        }

    # ---- roofline leg: per-kernel HIP-event timing of one extra batch (rank 0, outside the clock) --
    if rank == 0 and not args.no_roofline:
        _lib.profile_enable(True)
        reps = 3
        for _ in range(reps):
            net.score(batches[0][0], measure=args.measure)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        log("roofline leg done")
        dom = max(prof, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        sec = d["ms"] * 1e-3
        ai = d["flops"] / d["bytes"]
        ridge = FP32_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        if ai >= ridge:
            bound, achieved, peak, unit = "mfma", d["flops"] / sec / 1e12, FP32_PEAK_TFLOPS, "TFLOP/s"
        else:
            bound, achieved, peak, unit = "hbm", d["bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        total_ms = sum(v["ms"] for v in prof.values())
        # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
        # (tools/gpu_pmc.sh + tools/pmc_summary.py: separate --pmc runs; (2*FETCH_SIZE + WRITE_SIZE)*1024,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null if not collected
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc", "traffic_%s.json" % args.model)))
            traffic = pmc.get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        result["roofline"] = {
            "kernel": dom, "bound": bound, "achieved": achieved, "peak": peak, "unit": unit,
            "frac": achieved / peak, "traffic": traffic,
            "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
            "algorithmic_flops_per_launch": d["flops"] / d["launches"],
            "avg_launch_us": 1e3 * d["ms"] / d["launches"], "launches_per_batch": d["launches"] // reps,
            "share_of_gpu_time": d["ms"] / total_ms,
            "arithmetic_intensity_flop_per_byte": ai,
            "achieved_tflops": d["flops"] / sec / 1e12, "achieved_gbs": d["bytes"] / sec / 1e9,
            "per_kernel_ms_per_batch": {k: v["ms"] / reps for k, v in sorted(prof.items())},
        }

    # ---- CPU baseline leg (rank 0, N=1 only) ------------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        P = syn.enet_params_dict(net)
        c1 = None
        if args.model == "enet":
            net1 = ssal.ENet(19)
            net1.build((None, None, None, 3))
            syn.randomize_enet(net1, seed=0)
            c1 = (syn.enet_params_dict(net1), 19)
        result["cpu_baseline"] = cpu_baseline(P, h, w, c, args.measure, args.cpu_seconds, c1=c1, model=args.model)
        result["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]

    if rank == 0:
        # only meaningful when the whole pool was scored (default --steps 372 at N=1, or --scaling strong)
        # (and no resident batch was scored twice: --resident-gib did not make the batch list wrap around)
        whole_pool = int(total_frames) >= POOL and n_resident >= min(args.steps, n_batches_shard)
        result["top_k_checksum"] = int(np.sort(low).astype(np.int64).sum()) if whole_pool else None
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
