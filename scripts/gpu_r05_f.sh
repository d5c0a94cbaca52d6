#!/bin/bash
# Round 5: persistent form of gemm256 (256 workgroups walk the tile list): bit-identity, per-shape times, step A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
LIBV=${1:-persist}
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_a.txt 2> gpurun_out/hash_a.err; rc=$?; ok $rc || exit $rc
YAT_HIP_LIB=$V/libyat_$LIBV.so timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_b.txt 2> gpurun_out/hash_b.err; rc=$?; ok $rc || exit $rc
if diff -q gpurun_out/hash_a.txt gpurun_out/hash_b.txt > /dev/null; then echo "HASH IDENTICAL ($(wc -l < gpurun_out/hash_a.txt) lines)"; else echo "HASH DIFFERS"; diff gpurun_out/hash_a.txt gpurun_out/hash_b.txt | head -10; fi
rm -f gpurun_out/probe_persist.txt
export PROBE_CUSTOM="nn:8192:11200:1152,nn:8192:11200:2240,nn:8192:11200:4480,nn:8192:6720:2240,nt:8192:5600:2240,nn:32768:4608:1152,nn:32768:1152:4608,nt:32768:1152:1152,tt:11200:2240:8192"
for lib in product $LIBV product $LIBV; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  echo "== $lib" >> gpurun_out/probe_persist.txt
  timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_persist.txt 2> gpurun_out/probe_persist.err; rc=$?
  ok $rc || exit $rc
done
unset YAT_HIP_LIB PROBE_CUSTOM
cat gpurun_out/probe_persist.txt | cut -c1-160
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V/libyat_$LIBV.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_$LIBV.so" "YAT_X=0" 30
