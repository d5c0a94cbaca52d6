#!/bin/bash
# Round 5: which operand's coldness costs the forward GEMMs; the step's per-shape GEMM table on the new kernels.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
PROBE_AB=1 PROBE_SHAPES=3,4,5,6,7 timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/probe_ab.txt 2> gpurun_out/probe_ab.err; rc=$?
cat gpurun_out/probe_ab.txt; ok $rc || exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gemm-detail gpurun_out/gemm_per_shape.txt > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err; rc=$?
echo "bench rc=$rc"; tail -c 1500 gpurun_out/bench_b.json; head -n 30 gpurun_out/gemm_per_shape.txt
