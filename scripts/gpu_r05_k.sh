#!/bin/bash
# Round 5: cross-attention dK/dV with 128-key workgroups (8 waves) on a 128-key work list vs 64-key (4 waves): micro-benchmark
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_sdpatune.so
for r in 1 2; do
  echo "== 64-key work list (product policy)"; YAT_HIP_LIB=$V timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep sdpa || exit 1
  echo "== 128-key work list, 8 waves"; YAT_HIP_LIB=$V YAT_SDPA_WORK_KT128=1 timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep sdpa || exit 1
done
