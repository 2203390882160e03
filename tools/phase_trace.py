"""Per-wave phase timeline of the fused bottleneck kernels (measurement aid).

Builds libssal_hip_trace.so (= the product sources + -DSSAL_PHASE_TRACE), runs single ENet layers at
the bench shape (batch x 1024 x 2048 input => the layer's own resolution) and prints, per kernel,
how long the waves spend between the phase marks (shader clocks, s_memtime) and how the co-resident
workgroups of a CU overlap.  Usage:  python tools/phase_trace.py [--build-only] [--batch 8] [layer ...]
"""
import argparse, os, subprocess, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "semanticsegmentationactivelearning_amd")
TRACE_LIB = os.path.join(PKG, "libssal_hip_trace.so")
# _lib reads SSAL_LIB_PATH when the package is first imported (build_trace_lib imports it too): set it up front
os.environ["SSAL_LIB_PATH"] = TRACE_LIB


MEASURE_LIB = os.path.join(PKG, "libssal_hip_measure.so")


def build_trace_lib(measure_only=False):
    """-DSSAL_PHASE_TRACE: phase marks + ablation + env knobs (libssal_hip_trace.so); measure_only: -DSSAL_MEASURE alone
    (libssal_hip_measure.so: ablation + env knobs, no s_memtime marks in the kernels -- the library A/B timings use)"""
    from semanticsegmentationactivelearning_amd import build as B
    out, flag = (MEASURE_LIB, "-DSSAL_MEASURE") if measure_only else (TRACE_LIB, "-DSSAL_PHASE_TRACE")
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc"] + B.FLAGS + [flag, "-o", out] + B.sources(measure=True)
    print("[trace build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


LAYER_SHAPE = {  # name -> (H divisor, channels in)
    "Bottleneck1_1": (4, 64), "Bottleneck2_1": (8, 128), "Bottleneck2_2": (8, 128), "Bottleneck2_4": (8, 128),
    "Bottleneck2_6": (8, 128), "Bottleneck2_8": (8, 128), "Bottleneck5_1": (2, 16),
    "Bottleneck2_3": (8, 128), "Bottleneck2_0": (4, 64), "Bottleneck4_0": (8, 128),
}
MARKS = {
    "Bottleneck2": ["start", "proj done", "barrier", "conv mt0", "exp mt0", "conv mt1", "exp mt1", "stores acked"],
    "Bottleneck1": ["start", "centre proj", "ring+opnds", "barrier", "conv mt0", "store mt0", "all issued", "stores acked"],
    "Bottleneck5": ["start", "centre proj", "ring+opnds", "barrier", "conv mt0", "store mt0", "all issued", "stores acked"],
    "Bottleneck4": ["start", "ring proj", "centre proj", "barrier", "conv mt0", "store mt0", "all issued", "stores acked"],
}
MARKS_BY_LAYER = {"Bottleneck2_0": MARKS["Bottleneck4"],
                  "Bottleneck2_3": ["start", "proj done", "barrier", "5x1 half0", "barrier", "1x5+exp half0", "half1 done", "stores acked"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--build-measure", action="store_true", help="build libssal_hip_measure.so (-DSSAL_MEASURE) and exit")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("layers", nargs="*", default=["Bottleneck2_1", "Bottleneck1_1"])
    args = ap.parse_args()
    if args.build_measure:
        build_trace_lib(measure_only=True)
        return
    src_dir = os.path.join(PKG, "csrc")
    newest_src = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir) if not f.startswith("."))
    stale = os.path.exists(TRACE_LIB) and os.path.getmtime(TRACE_LIB) < newest_src
    if args.build_only or stale or not os.path.exists(TRACE_LIB):
        build_trace_lib()
        if args.build_only:
            return
    os.environ["SSAL_LIB_PATH"] = TRACE_LIB
    import numpy as np, torch
    from semanticsegmentationactivelearning_amd import _lib, models, synthetic
    L = _lib.lib()
    net = models.ENet(19)
    net.build((None, 1024, 2048, 3))
    synthetic.randomize_enet(net, seed=7)
    H, W, n = 1024, 2048, args.batch
    for name in args.layers:
        div, cin = LAYER_SHAPE[name]
        h, w = H // div, W // div
        x = torch.randn(n, h, w, cin, device="cuda")
        layer = [l for l in net.layers if l.name == name][0]
        amax = None
        if name == "Bottleneck4_0":  # an upsample block needs the pooling indices of its downsample twin
            down = [l for l in net.layers if l.name == "Bottleneck2_0"][0]
            _, amax = net._run_layer(down, torch.randn(n, 2 * h, 2 * w, 64, device="cuda"), None, True)
        for _ in range(2):
            net._run_layer(layer, x, amax, False)
        nbytes = 64 << 20
        buf = torch.zeros(nbytes // 8, dtype=torch.int64, device="cuda")
        _lib.check(L.ssal_debug_set_trace(_lib.dev_ptr(buf), nbytes))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); net._run_layer(layer, x, amax, False); e1.record()
        torch.cuda.synchronize()
        _lib.check(L.ssal_debug_set_trace(None, 0))
        t = buf.cpu().numpy().reshape(-1, 16)
        wg_of_row = np.arange(len(t)) // 4
        keep = t[:, 0] != 0
        t, wg_of_row = t[keep], wg_of_row[keep]
        marks = MARKS_BY_LAYER.get(name, MARKS[name[:11]])
        print("\n=== %s  [%d,%d,%d,%d]  %d waves traced, launch %.1f us (incl. host gaps)" % (name, n, h, w, cin, len(t), 1e3 * e0.elapsed_time(e1)))
        cyc = (t[:, 7] - t[:, 0]).astype(np.float64)
        rt = (t[:, 13] - t[:, 12]).astype(np.float64) * 10e-9  # 100 MHz
        ok = rt > 0
        print("wave lifetime: mean %.0f cycles (p10 %.0f, p90 %.0f) = %.1f us; shader clock %.2f GHz" % (
            cyc.mean(), np.percentile(cyc, 10), np.percentile(cyc, 90), 1e6 * rt.mean(), (cyc[ok] / rt[ok]).mean() / 1e9))
        print("kernel span (first mark .. last mark, realtime): %.1f us" % ((t[:, 13].max() - t[:, 12].min()) * 10e-3))
        for k in range(1, 8):
            dlt = (t[:, k] - t[:, k - 1]).astype(np.float64)
            print("  %-12s -> %-12s  mean %8.0f cyc  p10 %8.0f  p90 %8.0f   (%4.1f %% of lifetime)" % (
                marks[k - 1], marks[k], dlt.mean(), np.percentile(dlt, 10), np.percentile(dlt, 90), 100 * dlt.mean() / cyc.mean()))
        # timeline per CU: how many workgroup generations, and start offsets between co-resident WGs
        hw = t[:, 14]
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; xcc = t[:, 15] & 0xF
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ids, counts = np.unique(cuid, return_counts=True)
        print("distinct CUs seen: %d; waves per CU: min %d max %d" % (len(ids), counts.min(), counts.max()))
        start = t[:, 12].astype(np.float64); start -= start.min()
        one = np.sort(start[cuid == ids[0]][::1]) * 10e-3
        print("wave start times on one CU (us):", np.round(one[:48], 1).tolist())
        sel = cuid == ids[0]
        order = np.argsort(start[sel])
        print("  workgroup ids on that CU, by start time:", wg_of_row[sel][order][::4][:24].tolist())
        end = (t[:, 13].astype(np.float64) - t[:, 12].min()) * 10e-3
        for wg in wg_of_row[sel][order][::4][:12]:
            m = wg_of_row == wg
            ph = (t[m][:, 1:8] - t[m][:, 0:1]).astype(np.float64).mean(axis=0) / 2.1e3
            print("    wg %5d: start %6.1f us  end %6.1f us   marks (us after start, mean of 4 waves): %s" % (
                wg, 10e-3 * start[m].min(), end[m].max(), np.round(ph, 1).tolist()))
        sel = cuid == ids[len(ids) // 2]
        order = np.argsort(start[sel])
        print("  ... and on another CU:", wg_of_row[sel][order][::4][:24].tolist())


if __name__ == "__main__":
    main()
