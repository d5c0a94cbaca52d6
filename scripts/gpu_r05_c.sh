#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python scripts/corun_probe.py > gpurun_out/corun_probe.txt 2> gpurun_out/corun_probe.err; rc=$?
cat gpurun_out/corun_probe.txt; tail -n 5 gpurun_out/corun_probe.err
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=yat_amd/build/variants/libyat_r04.so" 30
