#!/bin/bash
# configs 3 / 4 with the current library and (same box) the two-stage GEMM schedule
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V0=yat_amd/build/variants/libyat_nodeep.so
for b in deep nodeep; do
  if [ $b = deep ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V0; fi
  timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 --gemm-detail gpurun_out/pixart_${b}_shapes.txt > gpurun_out/pixart_$b.json 2> gpurun_out/pixart_$b.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  timeout -k 10 400 python scripts/bench_sd35.py --steps 6 --warmup 2 --gemm-detail gpurun_out/sd35_${b}_shapes.txt > gpurun_out/sd35_$b.json 2> gpurun_out/sd35_$b.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python - <<PY
import json
for m in ("pixart", "sd35"):
    try:
        d = json.load(open("gpurun_out/%s_$b.json" % m)); r = d.get("roofline", {})
        print("$b %s: %.1f ms/step %.2f img/s  gemm serialized %.1f ms  %.0f TF/s  host %.1f ms" % (m, d["ms_per_step"], d["value"], r.get("gemm_ms_per_step_serialized", 0), r.get("achieved", 0), d.get("host_enqueue_ms_per_step", 0)))
    except Exception as e:
        print("$b", m, "failed", e)
PY
done
unset YAT_HIP_LIB
