#!/bin/bash
# Same-box A/B of two library builds on a side bench: gpu_ab_side.sh "scripts/bench_sd35.py --steps 6 --warmup 3 --roofline-steps 1" "ENV.." "ENV.." [rounds]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CMD="$1"; A="$2"; B="$3"; R="${4:-2}"
for r in $(seq 1 $R); do for setting in "$A" "$B"; do
  env $setting timeout -k 10 400 python $CMD > gpurun_out/ab.json 2> gpurun_out/ab.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('[%s] round %s: %.2f ms  loss %.6f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['loss']))" "$setting" "$r"
done; done
