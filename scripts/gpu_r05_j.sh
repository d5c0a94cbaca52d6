#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
rm -f gpurun_out/store_pattern.txt
for shape in "8192 11200" "8192 6720" "8192 2240" "32768 3456"; do
  for mode in 0 1 2 3 1 0; do
    timeout -k 5 60 scripts/probes/store_pattern $mode $shape 300 >> gpurun_out/store_pattern.txt 2>&1 || exit $?
  done
done
cat gpurun_out/store_pattern.txt
