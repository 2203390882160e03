"""Copy the evidence of the last tools/refresh_profiles.sh session from gpurun_out/ (scratch) into profiles/<round>_*
(tracked).  Usage: python tools/collect_profiles.py [round=r02]"""
import glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r05"
SRC, DST = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for model in ("enet", "icnet", "enet_bf16x3"):
    fs = glob.glob(os.path.join(SRC, "prof_final_%s" % model, "runc", "*_kernel_stats.csv"))
    if fs:
        # keep the kernel rows (drop torch helper kernels' very long names by truncating the name column)
        rows = open(max(fs, key=os.path.getmtime)).read().splitlines()
        open(os.path.join(DST, "%s_%s_kernel_stats.csv" % (RND, model)), "w").write("\n".join(r[:400] for r in rows) + "\n")
    fs = glob.glob(os.path.join(SRC, "prof_final_%s_groups2" % model, "runc", "*_kernel_stats.csv"))
    if fs:
        rows = open(max(fs, key=os.path.getmtime)).read().splitlines()
        open(os.path.join(DST, "%s_%s_kernel_stats_two_chains.csv" % (RND, model)), "w").write("\n".join(r[:400] for r in rows) + "\n")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), model, RND])
for src, dst in (("bench_enet_full.json", "%s_bench_enet_full_pool.json" % RND), ("bench_icnet.json", "%s_bench_icnet.json" % RND),
                 ("bench_enet_full_detail.json", "%s_bench_enet_full_pool_detail.json" % RND),
                 ("bench_icnet_detail.json", "%s_bench_icnet_detail.json" % RND)):
    if os.path.exists(os.path.join(SRC, src)):
        shutil.copy(os.path.join(SRC, src), os.path.join(DST, dst))
print(sorted(os.listdir(DST)))
