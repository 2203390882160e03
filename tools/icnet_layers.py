"""Per-layer time of one ICNet batch from a rocprofv3 kernel trace (gpurun_out/<dir>/runc/*_kernel_trace.csv):
maps the launches of the LAST traced batch, in launch order, to the ICNET_SPEC layers and prints achieved TFLOP/s.
Usage: python tools/icnet_layers.py gpurun_out/prof_ic2 [batch]"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import icnet_oracle as io

d = sys.argv[1]
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
f = max(glob.glob(os.path.join(d, "runc", "*_kernel_trace.csv")), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("void ssal::", "").replace("ssal::", ""),
       (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
starts = [i for i, k in enumerate(ks) if k[0].startswith("k_conv_first")]
start = starts[-2]
end = [i for i, k in enumerate(ks) if k[0].startswith("k_reduce_mean")][-1]
specs = {s[0]: s for s in io.conv_specs()}
seq = ks[start:end + 1]
names = [k for k, _ in seq]
fused_front = any(k.startswith("k_front2") for k in names)          # conv1_sub1 + conv2_sub1 in one launch (knob ic_front)
fused_blocks = sum(k.startswith("k_bottleneck_mfma") for k in names)  # conv2_2 / conv2_3 as one launch each (score path)
order = ["conv1_1_3x3_s2", "conv1_2_3x3", "conv1_3_3x3", "pool1"]
for name, cin, mid, cout, s, dd, proj in io.BNECKS:
    if name == "conv3_2":
        order.append("resize")
    if proj:
        order.append(name + "_1x1_proj")
    if fused_blocks and not proj and (cin, mid, cout, s) == (128, 32, 128, 1):
        order.append(name + " (fused block)")
        continue
    order += [name + "_1x1_reduce", name + "_3x3", name + "_1x1_increase"]
tail = ["conv5_4_k1"] + (["conv1_sub1 + conv2_sub1"] if fused_front else ["conv1_sub1", "conv2_sub1"]) + \
       ["conv3_sub1", "conv3_1_sub2_proj", "conv_sub4", "conv3_sub1_proj", "conv_sub2", "conv6_cls", "upscore", "reduce"]
nppm = len(seq) - len(order) - len(tail)
order += ["ppm_%d" % i for i in range(nppm)] + tail
FUSED = {"conv1_sub1 + conv2_sub1": ("conv1_sub1", "conv2_sub1")}
for name, cin, mid, cout, s, dd, proj in io.BNECKS:
    FUSED[name + " (fused block)"] = (name + "_1x1_reduce", name + "_3x3", name + "_1x1_increase")
DIV = {"conv1_1_3x3_s2": 4, "conv1_2_3x3": 4, "conv1_3_3x3": 4, "conv1_sub1": 2, "conv2_sub1": 4, "conv3_sub1": 8,
       "conv3_1_sub2_proj": 16, "conv_sub4": 16, "conv3_sub1_proj": 8, "conv_sub2": 8, "conv6_cls": 4, "conv5_4_k1": 32}
def layer_flops(nm):
    _, kk, cin, cout, s, dd = specs[nm]
    dv = DIV.get(nm) or (8 if nm.startswith("conv2_") else 16 if nm.startswith("conv3_1_") else 32)
    px = batch * (1024 // dv) * (2048 // dv)
    return 2.0 * px * kk * kk * cin * cout, "%dx%d %4d->%-4d s%d d%d px=%d" % (kk, kk, cin, cout, s, dd, px)


tot = totf = 0.0
for (k, us), nm in zip(seq, order):
    fl = ""
    parts = FUSED.get(nm, (nm,) if nm in specs else ())
    if parts:
        flops = sum(layer_flops(p)[0] for p in parts)
        totf += flops
        fl = "%6.1f TF  %s" % (flops / us / 1e6, layer_flops(parts[0])[1] if len(parts) == 1 else "%d layers" % len(parts))
    tot += us
    print("%-26s %-20s %8.1f us  %s" % (nm, k[:20], us, fl))
print("total %.1f us   conv flops %.1f GF -> %.1f TF over the whole batch" % (tot, totf / 1e9, totf / tot / 1e6))
