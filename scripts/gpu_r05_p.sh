#!/bin/bash
# Round 5: forward chains of the PixArt-Sigma / SD3.5 steps re-swept on the round's GEMM kernels, one box
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/chains_sweep.txt
for r in 1 2; do
  for c in 1 2 3; do
    timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 --chains $c > gpurun_out/px.json 2> gpurun_out/px.err; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    echo "pixart chains $c: $(python3 -c "import json; print(round(json.loads(open('gpurun_out/px.json').read().strip().splitlines()[-1])['ms_per_step'],2))") ms/step" | tee -a gpurun_out/chains_sweep.txt
  done
  for c in 1 2; do
    timeout -k 10 500 python scripts/bench_sd35.py --steps 6 --warmup 3 --chains $c > gpurun_out/sd.json 2> gpurun_out/sd.err; rc=$?; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    echo "sd35 chains $c: $(python3 -c "import json; print(round(json.loads(open('gpurun_out/sd.json').read().strip().splitlines()[-1])['ms_per_step'],2))") ms/step" | tee -a gpurun_out/chains_sweep.txt
  done
done
