"""What this box's memory system sustains for plain streaming (torch copy / fill / sum), to calibrate
the HBM roofline the fused kernels are priced against."""
import torch, time
def t(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 268, 1024, 4096):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    tc = t(lambda: y.copy_(x)); tf = t(lambda: y.fill_(1.0)); ts = t(lambda: x.sum())
    ta = t(lambda: torch.add(x, 1.0, out=y))
    print("%5d MB: copy %.2f TB/s (r+w)  add %.2f TB/s (r+w)  fill %.2f TB/s  sum %.2f TB/s" % (mb, 2 * n * 4 / tc / 1e12, 2 * n * 4 / ta / 1e12, n * 4 / tf / 1e12, n * 4 / ts / 1e12))
