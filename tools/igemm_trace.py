"""Per-wave phase totals of the implicit-GEMM convolution kernel (measurement aid; builds libssal_hip_trace.so =
product sources + -DSSAL_PHASE_TRACE).  Usage: python tools/igemm_trace.py [layer ...]   (ICNET_SPEC layer names)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import phase_trace as PT

SHAPES = {  # name: (k, cin, cout, stride, dil, div of the INPUT tensor, up2)
    "conv_sub2": (3, 128, 128, 1, 2, 16, True), "conv_sub4": (3, 256, 128, 1, 2, 32, True),
    "conv5_3x3": (3, 256, 256, 1, 4, 32, False), "conv5_reduce": (1, 1024, 256, 1, 1, 32, False),
    "conv4_3x3": (3, 128, 128, 1, 2, 32, False), "conv1_3": (3, 32, 64, 1, 1, 4, False),
    "conv1_2": (3, 32, 32, 1, 1, 4, False), "conv5_proj": (1, 512, 1024, 1, 1, 32, False),
}


def main():
    names = sys.argv[1:] or ["conv_sub2", "conv5_3x3", "conv1_3"]
    if not os.path.exists(PT.TRACE_LIB) or os.environ.get("REBUILD"):
        PT.build_trace_lib()
    os.environ["SSAL_LIB_PATH"] = PT.TRACE_LIB
    import numpy as np, torch
    from semanticsegmentationactivelearning_amd import _lib
    from semanticsegmentationactivelearning_amd.models.util import conv_ops as cops
    L = _lib.lib()
    for nm in names:
        k, cin, cout, s, d, div, up2 = SHAPES[nm]
        n, h, w = 8, 1024 // div, 2048 // div
        x = torch.randn(n, h, w, cin, device="cuda")
        ker = (np.random.default_rng(0).normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
        for _ in range(2):
            cops.conv_bn_act(x, ker, s, d, relu=True, upsample2x=up2)
        nbytes = 64 << 20
        buf = torch.zeros(nbytes // 8, dtype=torch.int64, device="cuda")
        _lib.check(L.ssal_debug_set_trace(_lib.dev_ptr(buf), nbytes))
        _lib.set_knob("ablate", int(os.environ.get("ABLATE", "0")))  # bit0 no MFMA, bit1 no global loads, bit2 no LDS writes
        cops.conv_bn_act(x, ker, s, d, relu=True, upsample2x=up2)
        _lib.set_knob("ablate", 0)
        torch.cuda.synchronize()
        _lib.check(L.ssal_debug_set_trace(None, 0))
        t = buf.cpu().numpy().reshape(-1, 16)
        t = t[t[:, 0] != 0].astype(np.float64)
        life = t[:, 8] - t[:, 0]
        loop = t[:, 7] - t[:, 1]
        rt = (t[:, 13] - t[:, 12]) * 10e-9
        clk = (life / rt).mean() / 1e9
        nch = t[0, 9]
        span = (t[:, 13].max() - t[:, 12].min()) * 10e-3
        print("\n=== %s  %dx%d %d->%d s%d d%d up2=%s  x=[%d,%d,%d]  %d waves, %d chunks, kernel span %.1f us, clock %.2f GHz"
              % (nm, k, k, cin, cout, s, d, up2, n, h, w, len(t), nch, span, clk))
        print("wave lifetime %.0f cycles; prologue %.0f; K loop %.0f (%.0f per chunk); epilogue %.0f"
              % (life.mean(), (t[:, 1] - t[:, 0]).mean(), loop.mean(), loop.mean() / nch, (t[:, 8] - t[:, 7]).mean()))
        for lab, col in (("issue global loads", 2), ("LDS reads + MFMAs", 3), ("wait for global loads", 4),
                         ("LDS writes", 5), ("barrier", 6)):
            print("  %-22s %8.0f cycles per chunk  (%4.1f %% of the loop)" % (lab, t[:, col].mean() / nch,
                                                                              100 * t[:, col].mean() / loop.mean()))


if __name__ == "__main__":
    main()
