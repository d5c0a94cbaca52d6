#!/bin/bash
# deep LDS-DMA schedule vs the two-stage one (-DYAT_GEMM_DEEP=0 variant): bit identity, GEMM tests, sustained hot / cold probe
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_nodeep.so
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_deep.txt 2> gpurun_out/gemm_hash_deep.err; rc=$?; echo "hash deep rc=$rc"; tail -2 gpurun_out/gemm_hash_deep.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
YAT_HIP_LIB=$V timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_nodeep.txt 2> gpurun_out/gemm_hash_nodeep.err; rc=$?; echo "hash nodeep rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
if diff gpurun_out/gemm_hash_deep.txt gpurun_out/gemm_hash_nodeep.txt > gpurun_out/gemm_hash_diff.txt; then echo "BIT-IDENTICAL"; else echo "DIFFERENT:"; head -20 gpurun_out/gemm_hash_diff.txt; fi
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or glu_backward" -p no:cacheprovider > gpurun_out/tests_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -3 gpurun_out/tests_gemm.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
for rep in 1 2; do
  timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/sustained_deep_$rep.txt 2>&1; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  YAT_HIP_LIB=$V timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/sustained_nodeep_$rep.txt 2>&1; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
grep -h "^[nt][nt] " gpurun_out/sustained_deep_1.txt | cut -c1-140
echo ---- nodeep
grep -h "^[nt][nt] " gpurun_out/sustained_nodeep_1.txt | cut -c1-140
