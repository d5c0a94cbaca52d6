#!/bin/bash
# Round 5: loss after n steps, LoRA / LoKr (B = 32), second-operand-pair form vs pre_add form: identical at step 1 (lora_B / w1 start
# at zero), then apart by what rounding once instead of twice does to a bf16 training run.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for ad in lora lokr; do
for n in 1 2 4 9; do
  for mode in pair pre; do
    flag=""; [ $mode = pre ] && flag="--lokr-pre-add"
    timeout -k 10 300 python bench.py --$ad 8 --batch 32 --steps $n --warmup 0 --no-cpu-baseline --no-gemm-timer --roofline-steps 0 $flag > gpurun_out/traj.json 2> gpurun_out/traj.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python3 -c "
import json; d=json.loads(open('gpurun_out/traj.json').read().strip().splitlines()[-1]); print('$ad steps $n $mode loss', d['loss'])" || { tail -n 5 gpurun_out/traj.err; exit 1; }
  done
done
done
