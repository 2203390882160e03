#!/bin/bash
# rocprofv3 per-symbol averages of the 128-channel bottleneck kernels over a short ENet bench run
export TMPDIR=/tmp
rm -rf gpurun_out/prof_bnk
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bnk -- python3 bench.py --steps ${STEPS:-24} --warmup 2 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/prof_bnk.log 2>&1 || exit $?
f=$(ls -t gpurun_out/prof_bnk/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "bottleneck_mfma" in n or "sample_mfma" in n:
        print("%-72s calls %4s avg %7.1f us" % (n.replace("void ssal::", "")[:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/prof_bnk.log | cut -c1-160
