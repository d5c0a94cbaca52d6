#!/bin/bash
# In-step GEMM times with operands pre-touched into the Infinity Cache (scripts/gemm_pretouch_diag.py), one bench per mode.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mode in ${MODES:-none ab b a}; do
  PRETOUCH=$mode timeout -k 10 300 python scripts/gemm_pretouch_diag.py --steps 6 --warmup 3 --roofline-steps 4 --no-cpu-baseline \
      --gemm-detail gpurun_out/pretouch_${mode}_shapes.txt > gpurun_out/pretouch_${mode}.json 2> gpurun_out/pretouch_${mode}.err
  rc=$?
  echo "mode $mode rc=$rc"
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/pretouch_${mode}.json"))
    r = d["roofline"]
    print("  step %.2f ms  gemm serialized %.2f ms/step  %.0f TF/s  launches %d" % (d["ms_per_step"], r["gemm_ms_per_step_serialized"], r["achieved"], r["launches"]))
except Exception as e:
    print("  parse failed", e)
PY
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "killed by timeout: stopping"; exit $rc; }
done
exit 0
