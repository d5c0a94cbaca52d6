#!/bin/bash
# One box, every bench, a list of environment settings: gpu_policy_sweep.sh "A=1 B=2" "A=0" ...  (ms per step per bench)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ms() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'])" "$1" 2>/dev/null || echo fail; }
for setting in "$@"; do
  line="[$setting]"
  env $setting timeout -k 10 300 python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/ps.json 2> gpurun_out/ps.err; rc=$?; line="$line sana $(ms gpurun_out/ps.json)"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
  env $setting timeout -k 10 300 python bench.py --lokr 8 --batch 32 --steps 8 --warmup 3 --no-cpu-baseline --no-gemm-timer > gpurun_out/ps.json 2> gpurun_out/ps.err; rc=$?; line="$line lokr32 $(ms gpurun_out/ps.json)"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
  env $setting timeout -k 10 300 python bench.py --lora 16 --batch 32 --steps 8 --warmup 3 --no-cpu-baseline --no-gemm-timer > gpurun_out/ps.json 2> gpurun_out/ps.err; rc=$?; line="$line lora32 $(ms gpurun_out/ps.json)"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
  env $setting timeout -k 10 300 python scripts/bench_pixart.py --steps 6 --warmup 2 --roofline-steps 1 > gpurun_out/ps.json 2> gpurun_out/ps.err; rc=$?; line="$line pixart $(ms gpurun_out/ps.json)"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
  env $setting timeout -k 10 300 python scripts/bench_sd35.py --steps 5 --warmup 2 --roofline-steps 1 > gpurun_out/ps.json 2> gpurun_out/ps.err; rc=$?; line="$line sd35 $(ms gpurun_out/ps.json)"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "$line killed"; exit $rc; }
  echo "$line" | tee -a gpurun_out/policy_sweep.txt
done
