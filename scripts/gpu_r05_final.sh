#!/bin/bash
# Round 5 final tree: the whole GPU suite, smoke, default bench (+ per-shape GEMM table), rocprofv3 default + serialized kernel
# stats, PMC passes (SQ, FETCH_SIZE, WRITE_SIZE), side benches of configs 3 / 4 / 5 with their kernel stats, soak.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
WHAT="${1:-all}"
if [ "$WHAT" = all ] || [ "$WHAT" = a ]; then
  bash scripts/gpu_round.sh "tests smoke" || exit $?
  timeout -k 10 600 python bench.py --steps 20 --warmup 5 --gemm-detail gpurun_out/gemm_per_shape.txt > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
  echo "bench rc=$rc"; tail -c 600 gpurun_out/bench.json; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  bash scripts/gpu_round.sh "prof profserial" || exit $?
  bash scripts/gpu_pmc.sh "sq fetch write" || exit $?
fi
if [ "$WHAT" = all ] || [ "$WHAT" = b ]; then
  bash scripts/gpu_models.sh "pixart sd35 lokr lora" || exit $?
  bash scripts/gpu_prof_models.sh "pixart sd35" > gpurun_out/prof_models.txt 2>&1 || exit $?
  bash scripts/gpu_prof_lokr.sh || exit $?
  timeout -k 10 400 python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-gemm-timer > gpurun_out/soak_1000.json 2> gpurun_out/soak_1000.err; rc=$?
  echo "soak rc=$rc"; tail -c 400 gpurun_out/soak_1000.json
fi
