#!/bin/bash
# Round 5: PixArt-Sigma / SD3.5 steps with every stream joined (YAT_SERIAL=1) under rocprofv3: each kernel's own duration
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp YAT_SERIAL=1
for m in pixart sd35; do
  rm -rf gpurun_out/prof_serial_$m
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial_$m -o prof -- python3 scripts/bench_$m.py --steps 4 --warmup 2 --roofline-steps 1 > gpurun_out/prof_serial_$m.json 2> gpurun_out/prof_serial_$m.err; rc=$?
  echo "prof serial $m rc=$rc"; [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
