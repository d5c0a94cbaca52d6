#!/bin/bash
# Round 5: K rotation of the gemm256 loops (workgroups sharing a B panel start at different depths): kernel tests, hot / cold
# probe per build, step A/B.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_krot.txt 2> gpurun_out/hash_krot.err; rc=$?; echo "hash rc=$rc wrong=$(grep -c WRONG gpurun_out/hash_krot.txt)"; ok $rc || exit $rc
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/gemm_tests.log; ok $rc || exit $rc
rm -f gpurun_out/probe_krot.txt
for lib in product krot1 krot2 krot8 krot4d2 product krot1; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  echo "== $lib" >> gpurun_out/probe_krot.txt
  PROBE_AB=1 PROBE_SHAPES=3,4,5,6,7,0 timeout -k 10 300 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_krot.txt 2> gpurun_out/probe_krot.err; rc=$?
  ok $rc || exit $rc
done
unset YAT_HIP_LIB
cat gpurun_out/probe_krot.txt
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V/libyat_krot1.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_krot2.so" "YAT_HIP_LIB=$V/libyat_krot4d2.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_krot1.so" "YAT_X=0" 30
