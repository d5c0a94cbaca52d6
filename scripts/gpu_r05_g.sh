#!/bin/bash
# Round 5: is the GEMM family power-bound?  Same launches on random and on all-zero operands (same instruction stream, same
# cycles; zero operands do not toggle the matrix pipe), sustained.  Then the new ddp test.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export PROBE_SHAPES=3,5,6,7,2
echo "== random operands" > gpurun_out/probe_power.txt
timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_power.txt 2> gpurun_out/probe_power.err || exit $?
echo "== all-zero operands" >> gpurun_out/probe_power.txt
PROBE_ZERO=1 timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_power.txt 2>> gpurun_out/probe_power.err || exit $?
cat gpurun_out/probe_power.txt | cut -c1-170
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py tests/test_trainer_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/ddp_tests.log 2>&1; rc=$?
tail -n 5 gpurun_out/ddp_tests.log
exit $rc
