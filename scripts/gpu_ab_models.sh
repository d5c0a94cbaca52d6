#!/bin/bash
# A/B two environment settings on the PixArt and SD3.5 benches, alternating, N steps: gpu_ab_models.sh "A=1" "B=2" [steps]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
N="${3:-12}"
ms() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'])" "$1" 2>/dev/null || echo fail; }
for r in 1 2; do
  for setting in "$1" "$2"; do
    line="[$setting]"
    env $setting timeout -k 10 300 python scripts/bench_pixart.py --steps $N --warmup 3 --roofline-steps 1 > gpurun_out/pm.json 2> gpurun_out/pm.err; rc=$?; line="$line pixart $(ms gpurun_out/pm.json)"
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
    env $setting timeout -k 10 300 python scripts/bench_sd35.py --steps $N --warmup 3 --roofline-steps 1 > gpurun_out/pm.json 2> gpurun_out/pm.err; rc=$?; line="$line sd35 $(ms gpurun_out/pm.json)"
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
    echo "$line"
  done
done
