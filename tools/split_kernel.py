"""The split-operand (bf16x3) form of the regular 128-channel bottleneck against the exact-fp32 product kernel
(measurement library only: python tools/phase_trace.py --build-measure first).

  1. per layer (batch 8 x 128 x 256 x 128, the bench shape): HIP-event time per launch and max / RMS difference of the outputs
  2. a whole ENet forward on one 256 x 512 frame with the twelve regular 128-channel layers on the split kernel: max / RMS
     difference of the logits against the exact-fp32 forward (which the parity tests tie to the oracle bit for bit),
     label changes
Nothing here ships: the knob exists only in -DSSAL_MEASURE builds and bench.py refuses to time it as a result."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SSAL_LIB_PATH", os.path.join(ROOT, "semanticsegmentationactivelearning_amd", "libssal_hip_measure.so"))
import numpy as np
import torch

import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import _lib, synthetic as syn

net = ssal.ENet(19)
net.build((None, None, None, 3))
syn.randomize_enet(net, seed=0)


def timed(layer, x, reps=10):
    layer(x, training=False)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    try:
        for _ in range(reps):
            y = layer(x, training=False)
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
    finally:
        _lib.profile_enable(False)
    return y, {k: 1e3 * v["ms"] / v["launches"] for k, v in prof.items()}


x = torch.from_numpy(np.random.default_rng(5).normal(size=(8, 128, 256, 128)).astype(np.float32)).cuda()
FORMS = {1: "split   (8x32 tiles, traffic of the product kernel)", 2: "split_r (8x16 tiles, block input read once)"}
for name in ("Bottleneck2_1", "Bottleneck2_2", "Bottleneck2_4", "Bottleneck2_6", "Bottleneck2_8"):
    layer = getattr(net, name)
    _lib.set_knob("bnk_split", 0)
    ref, t0 = timed(layer, x)
    print("%-14s exact %s" % (name, {k.replace("k_", ""): round(v, 1) for k, v in t0.items()}))
    for form, label in FORMS.items():
        _lib.set_knob("bnk_split", form)
        got, t1 = timed(layer, x)
        _lib.set_knob("bnk_split", 0)
        d = (got.double() - ref.double())
        print("    %-52s %6.1f us | max |d| %.3e rms %.3e (|y| max %.2f rms %.3f)" % (
            label, list(t1.values())[0], d.abs().max().item(), d.pow(2).mean().sqrt().item(), ref.abs().max().item(),
            ref.pow(2).mean().sqrt().item()))

# ragged shape through both forms (border tiles, partial sub-images)
xr = torch.from_numpy(np.random.default_rng(6).normal(size=(2, 72, 136, 128)).astype(np.float32)).cuda()
for name in ("Bottleneck2_1", "Bottleneck2_4"):
    layer = getattr(net, name)
    _lib.set_knob("bnk_split", 0)
    ref = layer(xr, training=False).clone()
    for form in FORMS:
        _lib.set_knob("bnk_split", form)
        got = layer(xr, training=False)
        _lib.set_knob("bnk_split", 0)
        print("ragged 2x72x136 %-14s form %d: max |d| %.3e" % (name, form, (got.double() - ref.double()).abs().max().item()))

f = syn.synth_frames_device(7, 1, 256, 512, 3)
_lib.set_knob("bnk_split", 0)
ref = net(f, training=False).clone()
for form, label in FORMS.items():
    _lib.set_knob("bnk_split", form)
    got = net(f, training=False).clone()
    _lib.set_knob("bnk_split", 0)
    d = got.double() - ref.double()
    print("whole forward 256x512, 12 regular 128-channel layers on %s: logits max |d| %.3e rms %.3e (|logit| max %.1f); labels changed: %d of %d"
          % (label.split("(")[0].strip(), d.abs().max().item(), d.pow(2).mean().sqrt().item(), ref.abs().max().item(),
             int((got.argmax(-1) != ref.argmax(-1)).sum().item()), ref.shape[1] * ref.shape[2]))
