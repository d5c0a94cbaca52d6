#!/bin/bash
# Sensitivity of the SANA step to each non-GEMM kernel family: the bench with that family's launches skipped (scripts/step_ablation.py, WRONG
# results -- only ms/step is read), same box, interleaved with the full step.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ms() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'])" "$1" 2>/dev/null || echo fail; }
: > gpurun_out/step_ablation.txt
for r in 1 2; do
  for a in none dwfwd dwbwd dwfwd,dwbwd la ln sdpa gate adamw dwfwd,dwbwd,la,ln,sdpa,gate; do
    ABLATE=$([ $a = none ] && echo "" || echo $a) timeout -k 10 200 python scripts/step_ablation.py --steps 16 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/abl.json 2> gpurun_out/abl.err; rc=$?
    echo "round $r  skipped: $(printf %-34s $a)  $(ms gpurun_out/abl.json) ms/step" | tee -a gpurun_out/step_ablation.txt
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  done
done
