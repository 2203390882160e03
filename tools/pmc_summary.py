"""Summarise the rocprofv3 --pmc passes (gpurun_out/pmc_<model>_*) into profiles/<round>_pmc/: the raw
counter_collection CSVs plus traffic_<model>.json = per-kernel average FETCH_SIZE / WRITE_SIZE per launch (KB)
and the SQ pipe counters.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950
FETCH_SIZE reports half the bytes of wide (16 B/lane) reads (MI355X_MICROARCH.md, HBM section).
Usage: python tools/pmc_summary.py [model=enet] [round=r02]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
MODEL = sys.argv[1] if len(sys.argv) > 1 else "enet"
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r03"
DST = os.path.join(ROOT, "profiles", "%s_pmc" % ROUND)
os.makedirs(DST, exist_ok=True)


def short(name):
    """kernel names as ssal_profile_collect (bench.py's roofline leg) reports them"""
    n = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("ssal::", "")
    if n.startswith("k_front2"):  # k_front2<3, float, 2> -> k_front2<s2> (last template argument = stride of the second convolution)
        return "k_front2<s%s>" % n.rstrip(">").split(",")[-1].strip()
    if n.startswith("k_bottleneck_bf16x3") or n.startswith("k_bottleneck_asym_bf16x3"):
        return n.split("<")[0]
    if n.startswith("k_bottleneck_mfma<"):  # k_bottleneck_mfma<32, 2> (tile width, epilogue form) -> the launcher's profile name k_bottleneck_mfma<32>
        return "k_bottleneck_mfma<%s>" % n[n.index("<") + 1:n.rindex(">")].split(",")[0].strip()
    if n.startswith("k_bottleneck16") or n.startswith("k_bottleneck_mfma"):
        return n.replace(" ", "")
    if n.startswith("k_final_score"):  # k_final_score<19, false, true> = Bottleneck5_1 evaluated inside (bench: "k_final_score<fused 5_1>")
        return "k_final_score<fused 5_1>" if n.replace(" ", "").endswith(",true>") else "k_final_score"
    if n.startswith("k_igemm"):  # k_igemm<NT, UP2 (0 / 1 / 2), DUAL> -> the launchers' profile names k_igemm<4>, <4,up2>, <4,dual>
        args = [t.strip() for t in n[n.index("<") + 1:n.rindex(">")].split(",")]
        up2 = len(args) > 1 and args[1] in ("1", "2", "true")
        dual = len(args) > 2 and args[2] == "true"
        return "k_igemm<%s%s%s>" % (args[0], ",up2" if up2 else "", ",dual" if dual else "")
    return n.split("<")[0]


out = collections.defaultdict(dict)
for pas, dst in (("sq", "sq_counters.csv"), ("fetch", "fetch_size.csv"), ("write", "write_size.csv"), ("lds", "lds_counters.csv")):
    fs = sorted(glob.glob(os.path.join(SRC, "pmc_%s_%s" % (MODEL, pas), "runc", "*_counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    shutil.copy(fs[-1], os.path.join(DST, MODEL + "_" + dst))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not k.startswith("k_"):
            continue
        for c, v in cs.items():
            out[k][c] = sum(v) / len(v)
            out[k]["launches_sampled"] = len(v)
for k, d in out.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
json.dump(out, open(os.path.join(DST, "traffic_%s.json" % MODEL), "w"), indent=1, sort_keys=True)
for k in sorted(out):
    d = out[k]
    print("%-28s fetch %8.0f KB  write %8.0f KB  hbm/launch %7.1f MB  mfma_busy %.3g" % (
        k, d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0), d.get("hbm_bytes_per_launch", 0) / 1e6,
        d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
