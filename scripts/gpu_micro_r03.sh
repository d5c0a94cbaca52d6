#!/bin/bash
# Round-3 micro-benchmark evidence in one GPU-box session (each step under its own timeout, stop after a kill):
# row kernels, depthwise forward band vs streaming (tuning build), AdamW variants (tuning build).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T=yat_amd/build/variants/libyat_tune.so
run() { out=$1; shift; timeout -k 10 300 "$@" >> gpurun_out/$out 2>/dev/null; rc=$?; [ $rc -ne 124 ] && [ $rc -ne 137 ] || { echo "killed: $*"; exit $rc; }; }
: > gpurun_out/rowops_bench.txt; run rowops_bench.txt python scripts/rowops_bench.py
: > gpurun_out/dwconv_stream_ab.txt
for r in 1 2; do for v in 0 1; do
  echo "== YAT_DW_STREAM=$v round $r" >> gpurun_out/dwconv_stream_ab.txt
  run dwconv_stream_ab.txt env YAT_HIP_LIB=$T YAT_DW_STREAM=$v python scripts/dwconv_bench.py
done; done
for rows in 4 16 32; do
  echo "== YAT_DW_STREAM=1 YAT_DW_STREAM_ROWS=$rows" >> gpurun_out/dwconv_stream_ab.txt
  run dwconv_stream_ab.txt env YAT_HIP_LIB=$T YAT_DW_STREAM_ROWS=$rows python scripts/dwconv_bench.py
done
: > gpurun_out/adamw_variants.txt
for r in 1 2; do for v in 0 1 2 3; do run adamw_variants.txt env YAT_HIP_LIB=$T YAT_ADAMW_VARIANT=$v python scripts/adamw_bench.py; done; done
tail -4 gpurun_out/rowops_bench.txt; grep "32x32: in-step" gpurun_out/dwconv_stream_ab.txt; cat gpurun_out/adamw_variants.txt
