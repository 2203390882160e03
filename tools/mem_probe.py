"""What the memory system sustains for the access shapes of the fused kernels (copy y = x of an
[8,256,512,64] fp32 tensor = the stage-1 bottleneck's traffic), vs a linear copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the probe kernels live in the measurement library only (python tools/phase_trace.py --build-measure)
os.environ.setdefault("SSAL_LIB_PATH", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      "semanticsegmentationactivelearning_amd", "libssal_hip_measure.so"))
import torch
from semanticsegmentationactivelearning_amd import _lib
L = _lib.lib()
n, h, w = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 256, 512
x = torch.randn(n, h, w, 64, device="cuda"); y = torch.empty_like(x); y2 = torch.empty_like(x)
names = ["linear", "tile 8x32 frag", "tile 8x32 coalesced", "tile 8x32 frag+halo", "tile 8x32 coalesced+halo",
         "tile 4x64 frag", "tile 2x128 frag", "tile 1x256 frag", "tile 16x16 frag", "tile 8x32 frag, group by group",
         "linear, 16 float4/thread batch", "linear, 16 float4/thread loop", "linear, 4 float4/thread batch",
         "linear, 2 float4/thread batch", "linear, 16 float4/thread batch, slab order",
         "persistent 8x32 tiles, next tile prefetched, 768 WGs", "persistent 8x16 tiles, prefetched, 1024 WGs",
         "persistent 8x32 tiles, prefetched, 512 WGs", "persistent 8x16 tiles, prefetched, 2048 WGs",
         "tile 8x32 frag, every 2nd first-round WG 12k cycles late", "same, 25k cycles late", "same, 50k cycles late",
         "tile 8x32 walked row by row (load row r+1, store row r)", "same, two rows per step"]
only = [int(a) for a in sys.argv[2:]]
def run(mode, spin, reps=20, out=y):
    _lib.check(L.ssal_debug_copy_probe(mode, _lib.dev_ptr(x), _lib.dev_ptr(out), n, h, w, spin, _lib.stream_ptr()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(L.ssal_debug_copy_probe(mode, _lib.dev_ptr(x), _lib.dev_ptr(out), n, h, w, spin, _lib.stream_ptr()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("tensor [%d,%d,%d,64] fp32 = %.0f MB in + the same out" % (n, h, w, x.numel() * 4 / 1e6))
for mode in range(len(names)):
    if only and mode not in only:
        continue
    y.zero_()
    us = run(mode, 0)
    assert torch.equal(x, y), names[mode]
    line = "%-22s %6.1f us  %.2f TB/s (r+w algorithmic)" % (names[mode], us, 2 * x.numel() * 4 / us / 1e6)
    for spin in (2000, 8000, 20000):
        line += "   spin %5d: %6.1f us" % (spin, run(mode, spin))
    print(line)
