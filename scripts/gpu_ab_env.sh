#!/bin/bash
# Same-box A/B of environment settings on one bench script: gpu_ab_env.sh "scripts/bench_pixart.py --steps 8 --warmup 3 --roofline-steps 1" "A=0" "A=1" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
CMD="$1"; shift
ms() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f ms  loss %.6f' % (d['ms_per_step'], d['loss']))" "$1" 2>/dev/null || echo fail; }
: > gpurun_out/ab_env.txt
for r in 1 2; do
  for setting in "$@"; do
    env $setting timeout -k 10 300 python $CMD > gpurun_out/ab.json 2> gpurun_out/ab.err; rc=$?
    echo "[$setting] round $r: $(ms gpurun_out/ab.json)" | tee -a gpurun_out/ab_env.txt
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  done
done
