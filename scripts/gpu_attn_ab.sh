#!/bin/bash
# A/B of tuning switches on the attention micro-benchmark with the -DYAT_TUNING variant of sdpa.hip (scripts/build_variant.py
# tune sdpa.hip -DYAT_TUNING): gpu_attn_ab.sh "YAT_SDPA_WIDE4=0" "YAT_SDPA_WIDE4=1" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_tune.so
: > gpurun_out/attn_ab.txt
for r in 1 2; do
  for setting in "$@"; do
    echo "== [$setting] round $r" >> gpurun_out/attn_ab.txt
    env YAT_HIP_LIB=$V $setting timeout -k 10 200 python scripts/attn_bench_pixart.py 2>/dev/null | grep "no bias\|N=T=4096:" | grep -v "rel err" >> gpurun_out/attn_ab.txt; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  done
done
cat gpurun_out/attn_ab.txt
