#!/bin/bash
# Round 5: dK/dV kernel of the cross-attention with four query-tile stages (counted vmcnt) vs two: tests, micro-benchmark, step A/B
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_dkv2.so
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_sana_gpu.py tests/test_packed_text_gpu.py -m gpu -q -x -p no:cacheprovider -k "sdpa or attn or attention or sana or packed" > gpurun_out/sdpa_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/sdpa_tests.log; [ $rc -ne 0 ] && exit $rc
for r in 1 2; do
  echo "== 4 stages (product)"; timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep sdpa || exit 1
  echo "== 2 stages"; YAT_HIP_LIB=$V timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep sdpa || exit 1
done
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V" "YAT_X=0" 30
