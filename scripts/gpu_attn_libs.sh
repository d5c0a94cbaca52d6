#!/bin/bash
# A/B of whole-library variants on the attention micro-benchmark: gpu_attn_libs.sh product noslp ...  ("product" = libyat_hip.so)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/attn_libs.txt
for r in 1 2; do
  for v in "$@"; do
    lib=$PWD/yat_amd/build/variants/libyat_$v.so
    [ "$v" = product ] && lib=$PWD/yat_amd/libyat_hip.so
    echo "== [$v] round $r" >> gpurun_out/attn_libs.txt
    env YAT_HIP_LIB=$lib timeout -k 10 200 python scripts/attn_bench_pixart.py 2>gpurun_out/attn_libs.err | grep "no bias\|cross T" | grep -v "rel err" >> gpurun_out/attn_libs.txt; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  done
done
cat gpurun_out/attn_libs.txt
