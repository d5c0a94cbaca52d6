#!/bin/bash
# Same-box A/B of several settings over the default bench: gpu_ab_multi.sh ROUNDS STEPS "ENV=.. ENV=.." "ENV=.." ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R="$1"; S="$2"; shift 2
: > gpurun_out/ab_multi.txt
for r in $(seq 1 $R); do for setting in "$@"; do
  env $setting timeout -k 10 300 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-gemm-timer > gpurun_out/ab.json 2> gpurun_out/ab.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('[%s] round %s: %.2f ms  loss %.6f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['loss']))" "$setting" "$r" | tee -a gpurun_out/ab_multi.txt
done; done
