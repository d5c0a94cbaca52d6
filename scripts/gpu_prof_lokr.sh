#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lokr -o prof -- python3 bench.py --lokr 8 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > gpurun_out/prof_lokr_bench.json 2> gpurun_out/prof_lokr.err; rc=$?
echo "prof lokr rc=$rc"; find gpurun_out/prof_lokr -name "*kernel_stats*" | head -2
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --gemm-detail gpurun_out/lokr_shapes.txt > gpurun_out/lokr_bench.json 2> gpurun_out/lokr_bench.err; rc=$?
echo "bench lokr rc=$rc"; python -c "
import json; d=json.load(open('gpurun_out/lokr_bench.json')); print(d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step'], d['roofline']['gemm_ms_per_step_serialized'], d['roofline']['launches'])"
