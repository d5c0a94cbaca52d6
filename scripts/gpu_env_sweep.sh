#!/bin/bash
# schedule switches re-measured on the current kernels, one box, two rounds
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/env_sweep.txt
for rep in 1 2; do
  for cfg in "" "YAT_FWD_CHAINS=3" "YAT_FWD_CHAINS=1" "YAT_GROUPED_WGRAD=1" "YAT_GROUP_BIG_WGRAD=1" "YAT_DEFER_WGRAD=1" "YAT_GROUP_SMALL_WGRAD=0" "YAT_KEEP_GLU_U=0"; do
    env $cfg timeout -k 10 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-gemm-timer > gpurun_out/sweep.json 2> gpurun_out/sweep.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    echo "round $rep  $(printf %-26s "${cfg:-default}") $(python -c "import json; print('%.2f ms/step' % json.load(open('gpurun_out/sweep.json'))['ms_per_step'])" 2>/dev/null || echo fail)" | tee -a gpurun_out/env_sweep.txt
  done
done
