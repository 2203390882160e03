"""The opt-in SSAL_ARITH_BF16X3 kernels (csrc/ssal_bottleneck_bf16x3.hip) against the exact-fp32 product kernels, on the GPU:
  1. per layer at the bench shape (8 x 128 x 256 x 128): HIP-event time per launch, max / RMS difference of the outputs
  2. ragged shapes (border tiles, partial phase sub-images)
  3. a whole forward on one 256 x 512 frame: logits max / RMS difference, label changes; pooling indices equal
  4. the ranking pass on 6 batches of 8 full-size frames: images/s of both modes, per-image score differences
usage: python tools/bf16x3_check.py [--keep 0|1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import _lib, synthetic as syn

if "--keep" in sys.argv:
    _lib.set_knob("bf3_keep", int(sys.argv[sys.argv.index("--keep") + 1]))
print("knobs:", _lib.get_knobs())
net = ssal.ENet(19)
net.build((None, None, None, 3))
syn.randomize_enet(net, seed=0)


def timed(layer, x, arithmetic, reps=10):
    layer(x, training=False, arithmetic=arithmetic)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    try:
        for _ in range(reps):
            y = layer(x, training=False, arithmetic=arithmetic)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
    finally:
        _lib.profile_enable(False)
    return y, {k: 1e3 * v["ms"] / v["launches"] for k, v in prof.items()}


x = torch.from_numpy(np.random.default_rng(5).normal(size=(8, 128, 256, 128)).astype(np.float32)).cuda()
for name in ("Bottleneck2_1", "Bottleneck2_2", "Bottleneck2_3", "Bottleneck2_4", "Bottleneck2_6", "Bottleneck2_8"):
    layer = getattr(net, name)
    ref, t0 = timed(layer, x, "f32")
    got, t1 = timed(layer, x, "bf16x3")
    d = got.double() - ref.double()
    print("%-14s exact %s | bf16x3 %s | max |d| %.3e rms %.3e (|y| max %.2f rms %.3f)" % (
        name, {k.replace("k_", ""): round(v, 1) for k, v in t0.items()}, {k.replace("k_", ""): round(v, 1) for k, v in t1.items()},
        d.abs().max().item(), d.pow(2).mean().sqrt().item(), ref.abs().max().item(), ref.pow(2).mean().sqrt().item()), flush=True)

for shape in ((2, 72, 136, 128), (1, 8, 8, 128), (3, 17, 33, 128), (1, 40, 24, 128)):
    xr = torch.from_numpy(np.random.default_rng(6).normal(size=shape).astype(np.float32)).cuda()
    for name in ("Bottleneck2_1", "Bottleneck2_3", "Bottleneck2_4", "Bottleneck3_8"):
        layer = getattr(net, name)
        ref = layer(xr, training=False).clone()
        got = layer(xr, training=False, arithmetic="bf16x3")
        print("ragged %s %-14s max |d| %.3e" % (shape, name, (got.double() - ref.double()).abs().max().item()), flush=True)

f = syn.synth_frames_device(7, 1, 256, 512, 3)
ref = net(f, training=False).clone()
a_ref = [t.clone() for t in net.pooling_argmax()]
got = net(f, training=False, arithmetic="bf16x3").clone()
a_got = net.pooling_argmax()
d = got.double() - ref.double()
print("whole forward 256x512: logits max |d| %.3e rms %.3e (|logit| max %.1f); labels changed: %d of %d; pooling indices equal: %s"
      % (d.abs().max().item(), d.pow(2).mean().sqrt().item(), ref.abs().max().item(),
         int((got.argmax(-1) != ref.argmax(-1)).sum().item()), ref.shape[1] * ref.shape[2],
         all(torch.equal(p, q) for p, q in zip(a_ref, a_got))), flush=True)

batches = [syn.synth_frames_device(8 * b, 8, 1024, 2048, 3) for b in range(6)]
for mode in ("f32", "bf16x3", "f32", "bf16x3"):
    for xb in batches[:2]:
        net.score(xb, arithmetic=mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(5):
        out = [net.score(xb, arithmetic=mode) for xb in batches]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sc = torch.cat(out).cpu().numpy()
    if mode == "f32":
        base = sc
    print("ranking pass %-7s %.1f images/s; max |score - exact| %.3e" % (mode, 5 * 48 / dt, np.abs(sc - base).max()), flush=True)
