#!/bin/bash
# Round 5: dw backward pass 2 with s recomputed from z: bit identity against the build that reads s, times, dw tests, step A/B
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants/libyat_dwreads.so
timeout -k 10 200 python scripts/dw_s_from_z_check.py > gpurun_out/dw_a.txt 2> gpurun_out/dw_a.err || { tail -n 5 gpurun_out/dw_a.err; exit 1; }
YAT_HIP_LIB=$V timeout -k 10 200 python scripts/dw_s_from_z_check.py > gpurun_out/dw_b.txt 2> gpurun_out/dw_b.err || { tail -n 5 gpurun_out/dw_b.err; exit 1; }
echo "== s from z (product)"; cat gpurun_out/dw_a.txt; echo "== s read"; cat gpurun_out/dw_b.txt
if diff <(awk '{print $4}' gpurun_out/dw_a.txt) <(awk '{print $4}' gpurun_out/dw_b.txt) > /dev/null; then echo "DW HASH IDENTICAL"; else echo "DW HASH DIFFERS"; fi
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_sana_gpu.py -m gpu -q -x -p no:cacheprovider -k "dw or glu or sana or conv" > gpurun_out/dw_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/dw_tests.log; [ $rc -ne 0 ] && exit $rc
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V" "YAT_X=0" 30
