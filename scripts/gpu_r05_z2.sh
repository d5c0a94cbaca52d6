#!/bin/bash
# Round 5: LoKr projections behind each weight gradient instead of a serial tail: adapter tests, config-5 bench (twice)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_lokr_gpu.py tests/test_lora_gpu.py tests/test_fulldepth_gpu.py -m gpu -q -x -p no:cacheprovider -k "lokr or lora or adapter" > gpurun_out/lokr_proj_tests.log 2>&1; rc=$?
tail -n 6 gpurun_out/lokr_proj_tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer > gpurun_out/lokr_proj_$i.json 2> gpurun_out/lokr_proj_$i.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 -c "
import json; d=json.loads(open('gpurun_out/lokr_proj_$i.json').read().strip().splitlines()[-1]); print('run $i', d['ms_per_step'], d['value'], d['loss'], d['host_enqueue_ms_per_step'])" || { tail -n 5 gpurun_out/lokr_proj_$i.err; exit 1; }
done
timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer --lokr-pre-add > gpurun_out/lokr_proj_pre.json 2> gpurun_out/lokr_proj_pre.err
python3 -c "
import json; d=json.loads(open('gpurun_out/lokr_proj_pre.json').read().strip().splitlines()[-1]); print('pre_add form', d['ms_per_step'], d['value'], d['loss'])"
