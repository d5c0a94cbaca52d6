#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/prefetch_probe.py > gpurun_out/prefetch_probe.txt 2> gpurun_out/prefetch_probe.err; rc=$?
cat gpurun_out/prefetch_probe.txt; tail -n 3 gpurun_out/prefetch_probe.err; exit $rc
