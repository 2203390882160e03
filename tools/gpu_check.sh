#!/bin/bash
# One GPU-box session: smoke -> parity tests -> short bench -> rocprofv3 kernel stats.
# Stops (no further GPU step) if a step was killed / timed out; plain test failures do not stop it.
set -u
OUT=gpurun_out
mkdir -p $OUT
step() {  # step <name> <timeout_s> <cmd...>
    local name=$1 to=$2; shift 2
    echo "=== $name ===" | tee -a $OUT/summary.log
    timeout -k 10 "$to" "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc" | tee -a $OUT/summary.log
    tail -n 25 $OUT/$name.log
    if [ $rc -ge 124 ]; then echo "step $name was killed (rc=$rc): stopping" | tee -a $OUT/summary.log; exit $rc; fi
    return 0
}
: > $OUT/summary.log
STEPS=${BENCH_STEPS:-24}
step smoke 300 python __graft_entry__.py smoke
step pytest_gpu 900 python -m pytest tests -m gpu -q -p no:cacheprovider ${PYTEST_ARGS:-}
step bench 600 python bench.py --steps $STEPS --warmup 2
if [ "${SKIP_PROF:-0}" != "1" ]; then
    export TMPDIR=/tmp
    step rocprof 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline
    find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} sh -c 'echo "--- {}"; head -n 25 {}' | tee -a $OUT/summary.log
fi
echo "=== done ===" | tee -a $OUT/summary.log
