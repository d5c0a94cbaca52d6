#!/bin/bash
# Same-box A/B of the 128x128 kernel's per-round fixed cost in the tile policy (YAT_GEMM_FIXED_128, tuning build), every bench
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/policy128_ab.txt
: > $out
export YAT_HIP_LIB=$PWD/yat_amd/build/variants/libyat_tune128.so
run() {  # label, env, command...
  local label=$1 cfg=$2; shift 2
  env $cfg timeout -k 10 400 "$@" > gpurun_out/sweep.json 2> gpurun_out/sweep.err; local rc=$?
  { [ $rc -eq 124 ] || [ $rc -eq 137 ]; } && exit $rc
  echo "$label $(printf %-28s "${cfg:-fixed=0}") $(python -c "import json; print('%.2f ms/step' % json.load(open('gpurun_out/sweep.json'))['ms_per_step'])" 2>/dev/null || echo fail)" | tee -a $out
}
for rep in 1 2; do
  for cfg in "" "YAT_GEMM_FIXED_128=3e-6" "YAT_GEMM_FIXED_128=6e-6"; do
    run "r$rep lokr  " "$cfg" python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline
    run "r$rep lora  " "$cfg" python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline
    run "r$rep sana  " "$cfg" python bench.py --steps 20 --warmup 5 --no-cpu-baseline
    run "r$rep pixart" "$cfg" python scripts/bench_pixart.py --steps 8 --warmup 3
    run "r$rep sd35  " "$cfg" python scripts/bench_sd35.py --steps 6 --warmup 2
  done
done
