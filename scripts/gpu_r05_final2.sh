#!/bin/bash
# Round 5, last tree: whole GPU suite, smoke, bench (+ per-shape table), serialized + default kernel stats, side benches
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/gpu_round.sh "tests smoke" || exit $?
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --gemm-detail gpurun_out/gemm_per_shape.txt > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
echo "bench rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
bash scripts/gpu_round.sh "prof profserial" || exit $?
bash scripts/gpu_models.sh "pixart sd35 lokr lora" || exit $?
