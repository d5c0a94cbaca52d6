#!/bin/bash
# A/B two environment settings of bench.py on one box: gpu_ab.sh "VAR=a" "VAR=b" [steps]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
STEPS="${3:-40}"
i=0
for setting in "$1" "$2"; do
  i=$((i+1))
  env $setting timeout -k 10 400 python bench.py --steps "$STEPS" --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err; rc=$?
  grep -h "host enqueue of" gpurun_out/ab_$i.err; echo "[$setting] rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab_$i.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['loss'])" 2>&1 | tail -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed by timeout: stopping"; exit $rc; fi
done
exit 0
