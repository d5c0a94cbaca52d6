#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 200 python scripts/dw_s_from_z_check.py > gpurun_out/dw_c.txt 2> gpurun_out/dw_c.err || { tail -n 5 gpurun_out/dw_c.err; exit 1; }
cat gpurun_out/dw_c.txt
bash scripts/gpu_round.sh "tests smoke bench prof profserial" || exit $?
