"""What the fp32 matrix pipe of this device sustains (bare MFMA loops), vs the 157.3 TFLOP/s spec."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the probe kernels live in the measurement library only (python tools/phase_trace.py --build-measure)
os.environ.setdefault("SSAL_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      "semanticsegmentationactivelearning_amd", "libssal_hip_measure.so"))
import torch
from semanticsegmentationactivelearning_amd import _lib
L = _lib.lib()
out = torch.zeros(4096 * 256, device="cuda")
for shape in (32, 132, 232, 332, 432, 16, 516, 616):
    for blocks in (256, 512, 768):
        iters = 40000 if shape in (16, 516, 616) else 20000
        _lib.check(L.ssal_debug_mfma_peak(shape, blocks, 1000, _lib.dev_ptr(out), _lib.stream_ptr()))  # warm
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        _lib.check(L.ssal_debug_mfma_peak(shape, blocks, iters, _lib.dev_ptr(out), _lib.stream_ptr()))
        torch.cuda.synchronize()
        p = _lib.profile_collect(); _lib.profile_enable(False)
        for k, v in p.items():
            print("%s blocks=%d (%.1f waves/SIMD): %.1f TFLOP/s  (%.2f ms)" % (k, blocks, blocks * 4 / 1024.0, v["flops"] / v["ms"] / 1e9, v["ms"]))
