#!/bin/bash
# same-box A/B of library variants (yat_amd/build/variants/libyat_<name>.so; "product" = the in-tree library):
#   VARIANTS="product gap1 gap4" PROBE_SHAPES=0,1,2,6,7,8 REPS=2 bash scripts/gpu_ab_variants.sh
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for v in ${VARIANTS:-product}; do
    if [ $v = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=yat_amd/build/variants/libyat_$v.so; fi
    if [ -n "${PROBE_SHAPES:-}" ] && [ $rep = 1 ]; then
      PROBE_TIME_S=0.4 PROBE_WARM_S=0.6 timeout -k 10 300 python scripts/gemm_sustained_probe.py > gpurun_out/abv_probe_$v.txt 2>&1
      echo "== $v"; grep -h "^[nt][nt] " gpurun_out/abv_probe_$v.txt | cut -c1-112
    fi
    timeout -k 10 300 python bench.py --steps ${STEPS:-30} --warmup 8 --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/abv_${v}_${rep}.json 2> gpurun_out/abv_${v}_${rep}.err; rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python - <<PY
import json
d = json.load(open("gpurun_out/abv_${v}_${rep}.json")); r = d.get("roofline", {})
print("$v rep $rep: step %.2f ms  %.1f img/s  gemm serialized %.2f ms/step %.0f TF/s" % (d["ms_per_step"], d["value"], r.get("gemm_ms_per_step_serialized", 0), r.get("achieved", 0)))
PY
  done
done
unset YAT_HIP_LIB
