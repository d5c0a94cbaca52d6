#!/bin/bash
# Round 5: workgroup width of the cross-attention forward / dQ kernels (64- vs 128-query workgroups), tuning build
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_sdpatune.so
for r in 1 2; do
  for w in -1 0 1; do
    echo "== YAT_SDPA_WIDE=$w"
    YAT_HIP_LIB=$V YAT_SDPA_WIDE=$w timeout -k 10 200 python scripts/attn_bench.py 2>&1 | grep "sdpa" | cut -c1-140 || exit 1
  done
done | tee gpurun_out/sdpa_wide.txt
