import os, subprocess, sys, threading, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
os.environ["SSAL_LIB_PATH"]=os.path.join(ROOT,"semanticsegmentationactivelearning_amd","libssal_hip_measure.so")
import numpy as np, torch
import semanticsegmentationactivelearning_amd as ssal
from semanticsegmentationactivelearning_amd import synthetic as syn, _lib
net = ssal.ENet(19); net.build((None,None,None,3)); syn.randomize_enet(net, seed=0)
x = torch.from_numpy(np.random.default_rng(1).normal(size=(8,128,256,128)).astype(np.float32)).cuda()
layer = net.Bottleneck2_1
def smi():
    out = subprocess.run(["rocm-smi","--showclocks","--showpower"],capture_output=True,text=True).stdout
    sclk=[l for l in out.splitlines() if "sclk" in l]; pw=[l for l in out.splitlines() if "Power (W)" in l]
    return sclk[0].split("(")[-1].rstrip(")"), pw[0].split(":")[-1].strip()
for knob,label in ((0,"exact k_bottleneck_mfma<32>"),(1,"split (8x32, 3 WG/CU)"),(2,"split_r (input once)"),(3,"split<16> (4 WG/CU)")):
    _lib.set_knob("bnk_split", knob)
    for _ in range(3): layer(x, training=False)
    torch.cuda.synchronize()
    samples=[]
    def sampler():
        time.sleep(1.5)
        for _ in range(4):
            samples.append(smi()); time.sleep(0.25)
    th=threading.Thread(target=sampler); th.start()
    t0=time.perf_counter(); n=0
    while th.is_alive():
        for _ in range(200): layer(x, training=False)
        torch.cuda.synchronize(); n+=200
    dt=time.perf_counter()-t0
    print("%-30s %7.1f us/call  power %s W  sclk %s" % (label, 1e6*dt/n, " / ".join(s[1] for s in samples), samples[-1][0]))
_lib.set_knob("bnk_split", 0)
