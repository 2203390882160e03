"""Package power / reported shader clock while ONE layer runs in a loop (batch 8 at the bench shape): which kernels of the pass
sit at the 1400 W power cap?   python tools/clock_power_layers.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import synthetic as syn

net = ssal.ENet(19); net.build((None, None, None, 3)); syn.randomize_enet(net, seed=0)
rng = np.random.default_rng(1)
shapes = {"Bottleneck2_1": (8, 128, 256, 128), "Bottleneck2_3": (8, 128, 256, 128), "Bottleneck1_1": (8, 256, 512, 64),
          "Bottleneck2_0": (8, 256, 512, 64), "Bottleneck4_0": (8, 128, 256, 128), "Bottleneck5_1": (8, 512, 1024, 16)}


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = [l for l in out.splitlines() if "sclk" in l]
    pw = [l for l in out.splitlines() if "Power (W)" in l]
    return (sclk[0].split("(")[-1].rstrip(")") if sclk else "?"), (pw[0].split(":")[-1].strip() if pw else "?")


for name, shp in shapes.items():
    layer = getattr(net, name)
    x = torch.from_numpy(rng.normal(size=shp).astype(np.float32)).cuda()
    if name == "Bottleneck4_0":
        am = torch.zeros((8, 128, 256, 64), dtype=torch.int64, device="cuda")
        hw = torch.arange(128 * 256, device="cuda").view(1, 128, 256, 1)
        am = ((hw // 256) * 2 * 512 + (hw % 256) * 2) * 64 + torch.arange(64, device="cuda").view(1, 1, 1, 64)
        am = am.expand(8, -1, -1, -1).contiguous()
        call = lambda: layer(x, unpool_argmax=am, training=False)
    else:
        call = lambda: layer(x, training=False)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    stop = False
    samples = []

    def sampler():
        time.sleep(1.5)
        for _ in range(4):
            samples.append(smi())
            time.sleep(0.25)

    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while th.is_alive():
        for _ in range(200):
            call()
        torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
    print("%-14s %8.1f us/call   power %s W   sclk %s" % (name, 1e6 * dt / n, " / ".join(s[1] for s in samples), samples[-1][0]))
