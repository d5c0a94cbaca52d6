#!/bin/bash
# Round 5: LoKr (config 5, B = 32) step with every stream joined under rocprofv3: each kernel's own duration
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp YAT_SERIAL=1
rm -rf gpurun_out/prof_serial_lokr
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial_lokr -o prof -- python3 bench.py --lokr 8 --batch 32 --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > gpurun_out/prof_serial_lokr.json 2> gpurun_out/prof_serial_lokr.err; rc=$?
echo "prof serial lokr rc=$rc"; tail -c 300 gpurun_out/prof_serial_lokr.json
