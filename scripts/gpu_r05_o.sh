#!/bin/bash
# Round 5: is address translation what a cold GEMM operand costs?  TCP UTCL1 counters of the sustained probe, hot vs cold
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PROBE_WARM_S=0.05 PROBE_TIME_S=0.05
for mode in hot cold; do
  for shape in 3 2; do
    rm -rf gpurun_out/pmc_tlb_${mode}_$shape
    PROBE_MODES=$mode PROBE_SHAPES=$shape timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_tlb_${mode}_$shape -o pmc -- python3 scripts/gemm_sustained_probe.py > gpurun_out/pmc_tlb_${mode}_$shape.txt 2> gpurun_out/pmc_tlb_${mode}_$shape.err; rc=$?
    echo "$mode shape $shape rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python3 - $mode $shape <<'PY'
import csv, glob, sys
mode, shape = sys.argv[1:3]
f = glob.glob(f"gpurun_out/pmc_tlb_{mode}_{shape}/**/*counter_collection*.csv", recursive=True)
tot, n = {}, {}
for r in csv.DictReader(open(f[0])):
    if "gemm256" not in r["Kernel_Name"]:
        continue
    k = r["Counter_Name"]
    tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
print(mode, shape, {k: round(v / n[k]) for k, v in tot.items()}, "launches", max(n.values()) if n else 0)
PY
  done
done
