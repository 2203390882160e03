"""Experiment: does running two half-batches on two HIP streams (so that the kernels of one overlap the
memory / matrix phases of the other) beat one stream with the whole batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semanticsegmentationactivelearning_amd import models, synthetic
def make():
    net = models.ENet(19); net.build((None, 1024, 2048, 3)); synthetic.randomize_enet(net, seed=7); return net
nets = [make(), make()]
x = synthetic.synth_frames_device(0, 16, 1024, 2048, 3)
def run_single(b, iters):
    for _ in range(2): nets[0].score(x[:b])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): s = nets[0].score(x[:b])
    torch.cuda.synchronize(); return b * iters / (time.perf_counter() - t0), s
def run_dual(b, iters):
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [None, None]
    for k in range(2):
        with torch.cuda.stream(st[k]): nets[k].score(x[k * b:(k + 1) * b])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        for k in range(2):
            with torch.cuda.stream(st[k]): outs[k] = nets[k].score(x[k * b:(k + 1) * b])
    torch.cuda.synchronize(); return 2 * b * iters / (time.perf_counter() - t0), torch.cat(outs)
for b in (4, 8):
    r1, s1 = run_single(2 * b, 12)
    r2, s2 = run_dual(b, 12)
    print("batch %2d on one stream: %.1f img/s   |   2 streams x batch %d: %.1f img/s   identical scores: %s" % (2 * b, r1, b, r2, bool(torch.equal(s1, s2))))
