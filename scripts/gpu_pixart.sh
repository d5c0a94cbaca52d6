#!/bin/bash
# PixArt-Sigma side measurement on one GPU box: bench line + rocprof kernel stats (each step under its own timeout;
# nothing runs after a timeout kill).
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
STEPS="${1:-bench prof}"
rc=0
for s in $STEPS; do
  case $s in
    tests)
      timeout -k 10 600 python -m pytest tests/test_pixart_gpu.py -q -s -p no:cacheprovider > gpurun_out/pixart_tests.log 2>&1; rc=$?
      grep -E "passed|failed" gpurun_out/pixart_tests.log | tail -2 ;;
    bench)
      timeout -k 10 400 python scripts/bench_pixart.py --steps "${BENCH_STEPS:-8}" --warmup 3 --gemm-detail gpurun_out/pixart_gemm_detail.txt > gpurun_out/pixart_bench.json 2> gpurun_out/pixart_bench.err; rc=$?
      echo "bench rc=$rc"; tail -c 1500 gpurun_out/pixart_bench.json; tail -3 gpurun_out/pixart_bench.err ;;
    prof)
      export TMPDIR=/tmp
      timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pixart_prof -o prof -- python3 scripts/bench_pixart.py --steps 3 --warmup 2 --roofline-steps 1 > gpurun_out/pixart_prof.json 2> gpurun_out/pixart_prof.err; rc=$?
      echo "prof rc=$rc"; find gpurun_out/pixart_prof -name "*kernel_stats*" | head -3 ;;
  esac
  [ "$rc" -eq 124 ] || [ "$rc" -eq 137 ] && { echo "step $s killed by timeout (rc=$rc): stopping"; exit $rc; }
done
exit 0
