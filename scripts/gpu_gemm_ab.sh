#!/bin/bash
# Same-box A/B of the GEMM kernels: parity tests on the product library, then the six-shape timing (scripts/gemm_ablate.py)
# and the step bench, each with the product library and with a variant library (yat_amd/build/variants/libyat_$1.so).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V="yat_amd/build/variants/libyat_${1:-prev}.so"
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_sana_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gemm_tests.log
[ $rc -ne 0 ] && { echo "tests failed rc=$rc"; exit $rc; }
for r in 1 2; do
  echo "product: $(timeout -k 10 120 python scripts/gemm_ablate.py 2>&1 | tail -1)" | tee -a gpurun_out/gemm_ab.txt
  echo "variant: $(YAT_HIP_LIB=$V timeout -k 10 120 python scripts/gemm_ablate.py 2>&1 | tail -1)" | tee -a gpurun_out/gemm_ab.txt
done
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V" 30
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V" "YAT_X=0" 30
