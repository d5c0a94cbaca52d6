#!/bin/bash
# Round 5: soak (1000 steps) and the whole trainer fed from shards (300 steps) on the last tree
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-gemm-timer > gpurun_out/soak_1000.json 2> gpurun_out/soak_1000.err; rc=$?
echo "soak rc=$rc"; tail -c 300 gpurun_out/soak_1000.json; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 500 python bench.py --data shards --steps 300 --warmup 10 --no-cpu-baseline --no-gemm-timer > gpurun_out/trainer_shards_300.json 2> gpurun_out/trainer_shards_300.err; rc=$?
echo "trainer rc=$rc"; tail -c 300 gpurun_out/trainer_shards_300.json
