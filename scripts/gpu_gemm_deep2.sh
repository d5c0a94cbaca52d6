#!/bin/bash
# deep schedule: bit identity vs the two-stage build, then same-box step A/B (bench, both builds, twice) 
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_nodeep.so
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_deep.txt 2> gpurun_out/gemm_hash_deep.err; rc=$?; echo "hash deep rc=$rc"; tail -2 gpurun_out/gemm_hash_deep.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
YAT_HIP_LIB=$V timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_nodeep.txt 2> gpurun_out/gemm_hash_nodeep.err; rc=$?; echo "hash nodeep rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
if diff gpurun_out/gemm_hash_deep.txt gpurun_out/gemm_hash_nodeep.txt > gpurun_out/gemm_hash_diff.txt; then echo "BIT-IDENTICAL ($(wc -l < gpurun_out/gemm_hash_deep.txt) lines)"; else echo "DIFFERENT:"; head -20 gpurun_out/gemm_hash_diff.txt; fi
for rep in 1 2; do
  for b in deep nodeep; do
    if [ $b = deep ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --gemm-detail gpurun_out/ab_${b}_${rep}_shapes.txt > gpurun_out/ab_${b}_${rep}.json 2> gpurun_out/ab_${b}_${rep}.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_${b}_${rep}.json")); r = d["roofline"]
print("$b rep $rep: step %.2f ms  %.1f img/s  gemm serialized %.2f ms/step %.0f TF/s" % (d["ms_per_step"], d["value"], r["gemm_ms_per_step_serialized"], r["achieved"]))
PY
  done
done
unset YAT_HIP_LIB
