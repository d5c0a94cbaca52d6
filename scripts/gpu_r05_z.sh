#!/bin/bash
# Round 5: LoKr forward through the GEMM's second operand pair: adapter tests, then the config-5 bench with and without it
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lokr_gpu.py tests/test_lora_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/lokr_pair_tests.log 2>&1; rc=$?
tail -n 15 gpurun_out/lokr_pair_tests.log; [ $rc -ne 0 ] && exit $rc
for mode in pair pre pair pre; do
  flag=""; [ $mode = pre ] && flag="--lokr-pre-add"
  timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer $flag > gpurun_out/lokr_$mode.json 2> gpurun_out/lokr_$mode.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 -c "
import json; d=json.loads(open('gpurun_out/lokr_$mode.json').read().strip().splitlines()[-1]); print('$mode', d['ms_per_step'], d['value'], d['loss'], d['hbm_peak_gb'])" || { tail -n 5 gpurun_out/lokr_$mode.err; exit 1; }
done
