#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprof kernel stats.  Each GPU step runs under its own
# timeout; if a step is killed by its timeout (rc 124/137) nothing further is started.
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
STEPS="${1:-tests bench prof}"
ok_to_continue() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
rc=0
for s in $STEPS; do
  case $s in
    tests)
      timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest ${TEST_ARGS:-tests} -m gpu -q -s -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; rc=$?
      echo "exit=$rc" >> gpurun_out/tests_gpu.log; grep -E "passed|failed" gpurun_out/tests_gpu.log | tail -2 ;;
    benchsmall)
      timeout -k 10 180 python bench.py --layers 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err; rc=$?
      echo "benchsmall rc=$rc"; cat gpurun_out/bench_small.json; tail -8 gpurun_out/bench_small.err ;;
    bench)
      timeout -k 10 600 python bench.py --steps "${BENCH_STEPS:-20}" --warmup "${BENCH_WARMUP:-5}" > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?
      echo "bench rc=$rc"; tail -c 3000 gpurun_out/bench.json; tail -5 gpurun_out/bench.err ;;
    prof)
      export TMPDIR=/tmp
      timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timer > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; rc=$?
      echo "prof rc=$rc"; find gpurun_out/prof -name "*kernel_stats*" | head -3 ;;
    profserial)   # same bench with every stream joined: the per-kernel durations the roofline figure is taken from
      export TMPDIR=/tmp
      YAT_SERIAL=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timer > gpurun_out/prof_serial_bench.json 2> gpurun_out/prof_serial.err; rc=$?
      echo "profserial rc=$rc"; find gpurun_out/prof_serial -name "*kernel_stats*" | head -3 ;;
    smoke)
      timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
      tail -3 gpurun_out/smoke.log ;;
  esac
  ok_to_continue $rc || { echo "step $s killed by timeout (rc=$rc): stopping"; exit $rc; }
  [ "$s" = benchsmall ] && [ "$rc" -ne 0 ] && { echo "benchsmall failed: stopping"; exit $rc; }
done
exit 0
