#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_noepipre.so
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_a.txt 2> gpurun_out/gemm_hash_a.err; rc=$?; echo "hash product rc=$rc"; tail -1 gpurun_out/gemm_hash_a.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
YAT_HIP_LIB=$V timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_b.txt 2> gpurun_out/gemm_hash_b.err; rc=$?; echo "hash noepipre rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
if diff gpurun_out/gemm_hash_a.txt gpurun_out/gemm_hash_b.txt > gpurun_out/gemm_hash_diff.txt; then echo "BIT-IDENTICAL ($(wc -l < gpurun_out/gemm_hash_a.txt) lines)"; else echo "DIFFERENT:"; head -20 gpurun_out/gemm_hash_diff.txt; fi
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or glu_backward or act_bwd" -p no:cacheprovider > gpurun_out/tests_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -2 gpurun_out/tests_gemm.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
VARIANTS="product noepipre" REPS=2 STEPS=30 BENCH_ARGS="--gemm-detail gpurun_out/epipre_shapes.txt" bash scripts/gpu_ab_variants.sh
