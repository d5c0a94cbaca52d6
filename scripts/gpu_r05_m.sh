#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for st in 1 8; do
  echo "== prefetch reads every ${st}th 4-byte word"
  PREFETCH_STRIDE=$st timeout -k 10 300 python scripts/prefetch_probe.py 2>&1 | grep pair || exit 1
done | tee gpurun_out/prefetch_probe2.txt
