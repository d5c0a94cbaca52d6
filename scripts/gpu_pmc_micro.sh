#!/bin/bash
# PMC passes over a micro-benchmark script: gpu_pmc_micro.sh scripts/dwconv_bench.py
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
SCRIPT="$1"
run() { # name counters...
  name=$1; shift
  rm -rf gpurun_out/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o pmc -- python3 $SCRIPT > gpurun_out/pmc_$name.log 2> gpurun_out/pmc_$name.err
  rc=$?; echo "pmc $name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]
}
run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES || exit 1
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS || exit 1
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
python3 scripts/pmc_summary.py > gpurun_out/pmc_summary.txt 2>&1
python3 - <<'PY' >> gpurun_out/pmc_summary.txt
import csv, collections, glob, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_sq2/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
        out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD","SQ_INSTS_VMEM_WR","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_INST_CYCLES_VMEM","SQ_WAIT_INST_LDS"]
print("\nkernel".ljust(60), " ".join(c[-14:].rjust(14) for c in cols))
for k, c in out.items():
    print(k.ljust(60), " ".join(("%.3g" % (sum(c[x]) / len(c[x]))).rjust(14) if x in c else "-".rjust(14) for x in cols))
PY
cat gpurun_out/pmc_summary.txt
