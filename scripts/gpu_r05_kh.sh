#!/bin/bash
# Round 5, forward-layout GEMM with the DEEP schedule (half-major B image): bit-identity against the round-4 kernel, kernel
# tests, hot / cold sustained probe per build, step A/B.  Each GPU step under its own timeout; stop after a kill.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_product.txt 2> gpurun_out/hash_product.err; rc=$?; echo "hash product rc=$rc"; ok $rc || exit $rc
YAT_HIP_LIB=$V/libyat_r04.so timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_r04.txt 2> gpurun_out/hash_r04.err; rc=$?; echo "hash r04 rc=$rc"; ok $rc || exit $rc
if diff -q gpurun_out/hash_product.txt gpurun_out/hash_r04.txt > /dev/null; then echo "HASH IDENTICAL ($(wc -l < gpurun_out/hash_product.txt) lines)"; else echo "HASH DIFFERS"; diff gpurun_out/hash_product.txt gpurun_out/hash_r04.txt | head -20; fi
grep -c WRONG gpurun_out/hash_product.txt
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/gemm_tests.log; ok $rc || exit $rc
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
export PROBE_SHAPES=3,4,5,6,7,0
for lib in product r04 kh0 nopin product r04; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  echo "== $lib" | tee -a gpurun_out/probe_kh.txt
  timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_kh.txt 2> gpurun_out/probe_kh.err; rc=$?
  ok $rc || exit $rc
done
unset YAT_HIP_LIB
cat gpurun_out/probe_kh.txt
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V/libyat_r04.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_r04.so" "YAT_X=0" 30
