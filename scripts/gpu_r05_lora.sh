#!/bin/bash
# Round 5: plain LoRA forward through the GEMM's second operand pair: adapter tests, then the LoRA bench (B = 32) in both forms
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lora_gpu.py tests/test_lokr_gpu.py tests/test_fulldepth_gpu.py -m gpu -q -x -p no:cacheprovider -k "lokr or lora or adapter" > gpurun_out/lora_pair_tests.log 2>&1; rc=$?
tail -n 15 gpurun_out/lora_pair_tests.log; [ $rc -ne 0 ] && exit $rc
for mode in pair pre pair pre; do
  flag=""; [ $mode = pre ] && flag="--lokr-pre-add"
  timeout -k 10 400 python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer $flag > gpurun_out/lora_$mode.json 2> gpurun_out/lora_$mode.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 -c "
import json; d=json.loads(open('gpurun_out/lora_$mode.json').read().strip().splitlines()[-1]); print('$mode', round(d['ms_per_step'],2), round(d['value'],2), d['loss'], round(d['hbm_peak_gb'],1))" || { tail -n 5 gpurun_out/lora_$mode.err; exit 1; }
done
