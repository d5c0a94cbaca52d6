#!/bin/bash
# Same-box A/B of two builds of the library over the default bench: gpu_ab_lib.sh LIB_A LIB_B [rounds] [steps]
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
A="$1"; B="$2"; R="${3:-3}"; S="${4:-20}"
: > gpurun_out/ab_lib.txt
for r in $(seq 1 $R); do for L in "$A" "$B"; do
  YAT_HIP_LIB=$L timeout -k 10 300 python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-gemm-timer > gpurun_out/ab.json 2> gpurun_out/ab.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo killed; exit $rc; }
  python3 -c "import json,sys; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('[%s] round %s: %.2f ms  loss %.6f' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['loss']))" "$L" "$r" | tee -a gpurun_out/ab_lib.txt
done; done
