#!/bin/bash
# PixArt-Sigma / SD3.5 schedule switches re-measured on the current kernels, one box
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/env_sweep_models.txt
for rep in 1 2; do
  for cfg in "" "YAT_PIXART_CHAINS=1" "YAT_PIXART_CHAINS=3" "YAT_FUSE_ACT_BWD=1" "YAT_PIXART_SPLIT=0"; do
    env $cfg timeout -k 10 300 python scripts/bench_pixart.py --steps 8 --warmup 3 > gpurun_out/sweep.json 2> gpurun_out/sweep.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    echo "round $rep pixart $(printf %-24s "${cfg:-default}") $(python -c "import json; print('%.2f ms/step' % json.load(open('gpurun_out/sweep.json'))['ms_per_step'])" 2>/dev/null || echo fail)" | tee -a gpurun_out/env_sweep_models.txt
  done
  for cfg in "" "YAT_SD3_CHAINS=1" "YAT_SD3_CHAINS=3"; do
    env $cfg timeout -k 10 300 python scripts/bench_sd35.py --steps 6 --warmup 2 > gpurun_out/sweep.json 2> gpurun_out/sweep.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    echo "round $rep sd35   $(printf %-24s "${cfg:-default}") $(python -c "import json; print('%.2f ms/step' % json.load(open('gpurun_out/sweep.json'))['ms_per_step'])" 2>/dev/null || echo fail)" | tee -a gpurun_out/env_sweep_models.txt
  done
done
