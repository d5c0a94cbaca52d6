#!/bin/bash
# L2-miss traffic (FETCH_SIZE, gfx950 x2 correction) and time of three SANA GEMMs under tile-order variants of gemm256.hip:
#   for g in 1 2 8 16; do python scripts/build_variant.py g$g gemm256.hip -DYAT_GEMM_GROUP=$g; done;  gpu_gemm_traffic.sh "product g1 g2 g8 g16"
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/gemm_traffic_variants.txt
for v in ${1:-product}; do
  if [ "$v" = product ]; then L=""; else L="yat_amd/build/variants/libyat_$v.so"; fi
  echo "== $v (rows of 256-row tiles sharing a B panel in one XCD's run: ${v#g})" >> gpurun_out/gemm_traffic_variants.txt
  YAT_HIP_LIB=$L timeout -k 10 120 python scripts/gemm_traffic_probe.py >> gpurun_out/gemm_traffic_variants.txt 2>/dev/null; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  rm -rf gpurun_out/pmc_gt
  YAT_HIP_LIB=$L timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_gt -o pmc -- python3 scripts/gemm_traffic_probe.py > /dev/null 2> gpurun_out/pmc_gt.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 - >> gpurun_out/gemm_traffic_variants.txt <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_gt/**/*counter_collection*.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            name = re.sub(r"\(anonymous namespace\)::|void |\(GemmP\)", "", r["Kernel_Name"])
            agg[(name, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))].append(float(r["Counter_Value"]))
for (name, grid), v in sorted(agg.items()):
    print(f"   {name:34s} grid {grid:>8s}: L2-miss fetch {2.0 * 1024 * sum(v) / len(v) / 1e6:7.1f} MB per launch (FETCH_SIZE x 2, {len(v)} launches)")
PY
done
cat gpurun_out/gemm_traffic_variants.txt
