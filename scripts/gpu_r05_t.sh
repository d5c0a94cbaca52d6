#!/bin/bash
# Round 5: which pass-2 kernel per bucket after the s read went away (band kernel at 2 / 3 workgroups per CU, global-z kernel)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
for r in 1 2; do
  for cfg in "dwtune2 0" "dwtune2 2" "dwtune2 1" "dwtune3 2"; do
    set -- $cfg
    echo "== $1 YAT_DW_BWD2=$2 (0 policy, 2 band kernel, 1 global-z kernel)"
    YAT_HIP_LIB=$V/libyat_$1.so YAT_DW_BWD2=$2 timeout -k 10 200 python scripts/dw_s_from_z_check.py 2>&1 | grep "dw bwd2" || exit 1
  done
done | tee gpurun_out/dw_bwd2_kernels.txt
