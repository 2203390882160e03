for ab in 0 1 2; do
  SSAL_ABLATE=$ab SSAL_LIB_PATH=$(pwd)/semanticsegmentationactivelearning_amd/libssal_hip_trace.so python bench.py --allow-nondefault-knobs --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']['per_kernel_ms_per_batch']; print('ablate', $ab, {k: round(v,3) for k,v in r.items() if 'bottleneck' in k})"
done
