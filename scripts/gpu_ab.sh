#!/bin/bash
# Same-box A/B of any number of settings on one bench command (the one A/B driver; rounds 1-5 had a dozen special cases of it):
#   gpu_ab.sh [-r ROUNDS] "BENCH COMMAND" "ENV=.. ENV=.." "ENV=.." ...
# e.g. gpu_ab.sh "bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-gemm-timer" "X=0" "YAT_PRIO=chain=-1,side=1"
#      gpu_ab.sh -r 3 "scripts/bench_pixart.py --steps 8 --warmup 3 --roofline-steps 1" "YAT_HIP_LIB=yat_amd/libyat_hip.so" \
#                "YAT_HIP_LIB=yat_amd/build/variants/libyat_NAME.so"          (two builds of the library: scripts/build_variant.py)
# Settings alternate inside every round (box drift hits them alike); a run killed by its timeout stops the session.
# Results: gpurun_out/ab.txt (copy what is to be judged into profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
R=2
[ "${1:-}" = "-r" ] && { R="$2"; shift 2; }
CMD="$1"; shift
ms() { python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('%.2f ms  loss %.6f' % (d['ms_per_step'], d['loss']))" "$1" 2>/dev/null || echo fail; }
: > gpurun_out/ab.txt
for r in $(seq 1 "$R"); do
  for setting in "$@"; do
    env $setting timeout -k 10 "${AB_TIMEOUT:-400}" python $CMD > gpurun_out/ab.json 2> gpurun_out/ab.err; rc=$?
    echo "[$setting] round $r: $(ms gpurun_out/ab.json)" | tee -a gpurun_out/ab.txt
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "killed by timeout: stopping"; exit $rc; }
  done
done
exit 0
