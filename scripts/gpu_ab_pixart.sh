#!/bin/bash
# A/B two environment settings of scripts/bench_pixart.py on one box: gpu_ab_pixart.sh "VAR=a" "VAR=b" [steps]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
STEPS="${3:-8}"
i=0
for setting in "$1" "$2"; do
  i=$((i+1))
  env $setting timeout -k 10 400 python scripts/bench_pixart.py --steps "$STEPS" --warmup 3 > gpurun_out/abp_$i.json 2> gpurun_out/abp_$i.err; rc=$?
  echo "[pixart $setting] rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/abp_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'], d['roofline']['gemm_ms_per_step_serialized'])" 2>&1 | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed by timeout: stopping"; exit $rc; fi
done
exit 0
