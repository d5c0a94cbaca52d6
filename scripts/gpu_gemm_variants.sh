#!/bin/bash
# Same-box comparison of gemm256 variant libraries: gpu_gemm_variants.sh "name1 name2 ..." [rounds]; "product" = the in-tree library.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R="${2:-2}"
for r in $(seq 1 $R); do
  for n in $1; do
    if [ "$n" = product ]; then L=""; else L="yat_amd/build/variants/libyat_$n.so"; fi
    echo "$(printf %-8s $n): $(YAT_HIP_LIB=$L timeout -k 10 120 python scripts/gemm_ablate.py 2>&1 | tail -1)" | tee -a gpurun_out/gemm_variants.txt
  done
done
