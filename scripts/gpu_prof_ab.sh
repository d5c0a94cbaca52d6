#!/bin/bash
# rocprofv3 kernel stats of the bench with two libraries, default (overlapped) and serialized streams: which kernels move?
# gpu_prof_ab.sh VARIANT  ->  gpurun_out/pab_{product,VARIANT}_{ovl,ser}_stats.csv
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for n in product "$1"; do
  if [ "$n" = product ]; then export YAT_HIP_LIB=""; else export YAT_HIP_LIB="yat_amd/build/variants/libyat_$n.so"; fi
  for mode in ovl ser; do
    if [ $mode = ser ]; then export YAT_SERIAL=1; else unset YAT_SERIAL; fi
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pab_${n}_$mode -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timer > gpurun_out/pab_${n}_$mode.json 2> gpurun_out/pab_${n}_$mode.err; rc=$?
    echo "$n $mode rc=$rc $(python3 -c "import json; d=json.loads(open('gpurun_out/pab_${n}_$mode.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>&1 | tail -1)"
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    cp gpurun_out/pab_${n}_$mode/p_kernel_stats.csv gpurun_out/pab_${n}_${mode}_stats.csv 2>/dev/null || find gpurun_out/pab_${n}_$mode -name "*kernel_stats*" -exec cp {} gpurun_out/pab_${n}_${mode}_stats.csv \;
    rm -rf gpurun_out/pab_${n}_$mode
  done
done
exit 0
