#!/bin/bash
# rocprofv3 kernel stats of the PixArt-Sigma / SD3.5 side benches: gpu_prof_models.sh "pixart sd35"
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in ${1:-pixart sd35}; do
  rm -rf gpurun_out/prof_$m
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$m -o prof -- python3 scripts/bench_$m.py --steps 4 --warmup 2 --roofline-steps 1 > gpurun_out/prof_$m.json 2> gpurun_out/prof_$m.err; rc=$?
  echo "prof $m rc=$rc"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 - "$m" <<'PY'
import csv, glob, sys
m = sys.argv[1]
f = glob.glob(f"gpurun_out/prof_{m}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{m}: total kernel time {tot/1e6:.1f} ms over the profiled run")
for r in rows[:28]:
    print(f"  {r['Name'][:100]:100s} calls {r['Calls']:>6s}  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']:>6s} %")
PY
done
