#!/bin/bash
# Attention kernels at the PixArt-Sigma / SD3.5 shapes: micro-benchmark, parity tests of the attention kernels, then (optional
# "pmc") an SQ counter pass -- vector vs matrix busy -- and a kernel-stats pass over the same micro-benchmark.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
WHAT="${1:-bench tests}"
for s in $WHAT; do
  case $s in
    bench) timeout -k 10 300 python scripts/attn_bench_pixart.py > gpurun_out/attn_bench.txt 2> gpurun_out/attn_bench.err; rc=$?
           echo "attn bench rc=$rc"; cat gpurun_out/attn_bench.txt; tail -3 gpurun_out/attn_bench.err ;;
    tests) timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_packed_text_gpu.py -m gpu -q -s -p no:cacheprovider -k "sdpa or attention" > gpurun_out/attn_tests.log 2>&1; rc=$?
           echo "attn tests rc=$rc"; grep -E "passed|failed" gpurun_out/attn_tests.log | tail -2; grep -E "^FAILED|^E  " gpurun_out/attn_tests.log | head -20 ;;
    pmc)   rm -rf gpurun_out/attn_pmc gpurun_out/attn_stats
           timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/attn_pmc -o pmc -- python3 scripts/attn_bench_pixart.py > gpurun_out/attn_pmc.log 2> gpurun_out/attn_pmc.err; rc=$?
           echo "attn pmc rc=$rc"
           [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
           timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_stats -o stats -- python3 scripts/attn_bench_pixart.py > gpurun_out/attn_stats.log 2> gpurun_out/attn_stats.err; rc=$?
           echo "attn stats rc=$rc"
           python3 - <<'PY' > gpurun_out/attn_sq_summary.txt 2>&1
import csv, collections, glob, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/attn_pmc/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:64]
        if "sdpa" in name:
            out[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU"]
print("per-launch averages (SQ_WAVE_CYCLES / WAIT_* / ACTIVE_INST_* in quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES in cycles)")
print("kernel".ljust(66), " ".join(c.replace("SQ_", "")[-18:].rjust(18) for c in cols), "  valu_active/wave  mfma_busy/(4*wave)")
for k, c in sorted(out.items()):
    a = {x: (sum(c[x]) / len(c[x]) if x in c else float("nan")) for x in cols}
    print(k.ljust(66), " ".join(("%.4g" % a[x]).rjust(18) for x in cols),
          "  %.3f" % (a["SQ_ACTIVE_INST_VALU"] / a["SQ_WAVE_CYCLES"]), "  %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * a["SQ_WAVE_CYCLES"])))
PY
           cat gpurun_out/attn_sq_summary.txt; find gpurun_out/attn_stats -name "*kernel_stats*" | head -2 ;;
  esac
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "step $s killed by timeout: stopping"; exit $rc; }
done
exit 0
