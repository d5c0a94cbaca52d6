#!/bin/bash
# deep-nn schedule: bit identity of product vs both older schedules, sustained probe on the forward shapes, step A/B
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V0=yat_amd/build/variants/libyat_nodeep.so
V1=yat_amd/build/variants/libyat_nodeepnn.so
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_deep.txt 2> gpurun_out/gemm_hash_deep.err; rc=$?; echo "hash product rc=$rc"; tail -1 gpurun_out/gemm_hash_deep.txt
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
YAT_HIP_LIB=$V0 timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/gemm_hash_nodeep.txt 2> gpurun_out/gemm_hash_nodeep.err; rc=$?; echo "hash nodeep rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
if diff gpurun_out/gemm_hash_deep.txt gpurun_out/gemm_hash_nodeep.txt > gpurun_out/gemm_hash_diff.txt; then echo "BIT-IDENTICAL ($(wc -l < gpurun_out/gemm_hash_deep.txt) lines)"; else echo "DIFFERENT:"; head -20 gpurun_out/gemm_hash_diff.txt; fi
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm or glu_backward" -p no:cacheprovider > gpurun_out/tests_gemm.log 2>&1; rc=$?; echo "gemm tests rc=$rc"; tail -2 gpurun_out/tests_gemm.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
PROBE_SHAPES=3,4,5 timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/sustained_nn_deep.txt 2>&1
YAT_HIP_LIB=$V1 PROBE_SHAPES=3,4,5 timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/sustained_nn_nodeepnn.txt 2>&1
grep -h "^nn" gpurun_out/sustained_nn_deep.txt | cut -c1-150; echo "---- two-stage nn"; grep -h "^nn" gpurun_out/sustained_nn_nodeepnn.txt | cut -c1-150
for rep in 1 2; do
  for b in deep nodeepnn; do
    if [ $b = deep ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V1; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/ab_${b}_${rep}.json 2> gpurun_out/ab_${b}_${rep}.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python - <<PY
import json
d = json.load(open("gpurun_out/ab_${b}_${rep}.json")); r = d["roofline"]
print("$b rep $rep: step %.2f ms  %.1f img/s  gemm serialized %.2f ms/step %.0f TF/s" % (d["ms_per_step"], d["value"], r["gemm_ms_per_step_serialized"], r["achieved"]))
PY
  done
done
unset YAT_HIP_LIB
