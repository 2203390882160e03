"""PCIe-inclusive rate of the ranking pass: frames live in HOST memory (as the TFRecord front-end hands them
over) and are copied to the GPU batch by batch; float32 vs uint8 frames, with and without the side-stream
prefetch of active_learning.prefetch_to_device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from semanticsegmentationactivelearning_amd import models, synthetic, active_learning as al
net = models.ENet(19); net.build((None, 1024, 2048, 3)); synthetic.randomize_enet(net, seed=0)
nb, bs = 24, 8
u8 = [synthetic.synth_frames_device(b * bs, bs, 1024, 2048, 3, dtype=torch.uint8).cpu().numpy() for b in range(nb)]
f32 = [x.astype(np.float32) * np.float32(1 / 255.0) for x in u8]
def batches(frames):
    for b, x in enumerate(frames):
        yield x, np.arange(b * bs, (b + 1) * bs)
ref = None
u8_pinned = [torch.from_numpy(x).pin_memory() for x in u8]
f32_pinned = [torch.from_numpy(x).pin_memory() for x in f32[:12]] + f32[12:]
for name, frames in (("float32", f32), ("uint8", u8), ("f32 pinned(12)", f32_pinned[:12]), ("u8 pinned", u8_pinned)):
    for pf in (0, 2):
        al.rank_confidence(net, batches(frames[:2]), nb * bs, np.arange(nb * bs), 16, prefetch=pf)  # warm-up
        torch.cuda.synchronize(); t0 = time.perf_counter()
        low, uc = al.rank_confidence(net, batches(frames), nb * bs, np.arange(nb * bs), 16, prefetch=pf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None: ref = uc
        k = len(frames) * bs
        print("host %-14s frames, prefetch %d: %7.1f images/s   (identical scores: %s)" % (name, pf, k / dt, bool((uc[:k] == ref[:k]).all())))
