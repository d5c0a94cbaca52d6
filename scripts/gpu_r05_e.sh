#!/bin/bash
# Round 5: first-round stagger of gemm256 workgroups (epilogues out of lockstep): probe + step A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
rm -f gpurun_out/probe_stag.txt
for lib in product stag32 stag96 product; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  echo "== $lib" >> gpurun_out/probe_stag.txt
  PROBE_SHAPES=3,5,6,0,2 timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_stag.txt 2> gpurun_out/probe_stag.err; rc=$?
  ok $rc || exit $rc
done
unset YAT_HIP_LIB
cat gpurun_out/probe_stag.txt
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V/libyat_stag32.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_stag96.so" "YAT_X=0" 30
