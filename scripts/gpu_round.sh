#!/bin/bash
# One GPU-box session, the steps named on the command line in order (default: tests bench prof).  Each GPU step runs under
# its own timeout; if a step is killed by its timeout (rc 124/137) nothing further is started.
#   tests benchsmall bench prof profserial smoke     the round's standard evidence (parity, bench line, rocprofv3 kernel stats)
#   pixart sd35 lokr lora ddp                        side benches: configs 3 / 4 / 5 and the forced one-rank data-parallel line
#   soak shards                                      1000-step soak; the whole trainer fed from shards
#   ablation                                         the step with one non-GEMM kernel family skipped at a time (step_ablation.py)
# PMC passes: scripts/gpu_pmc.sh; same-box A/B of settings or library builds: scripts/gpu_ab.sh.  (The ~60 one-off session
# drivers of rounds 2-5 -- gpu_r05_*.sh and friends -- are in the git history up to commit 749ae92.)
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
    pixart) timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 --gemm-detail gpurun_out/pixart_gemm_per_shape.txt > gpurun_out/pixart_bench.json 2> gpurun_out/pixart_bench.err; rc=$?
            echo "pixart rc=$rc"; tail -c 1500 gpurun_out/pixart_bench.json; grep "host enqueue" gpurun_out/pixart_bench.err ;;
    sd35)   timeout -k 10 500 python scripts/bench_sd35.py --steps 6 --warmup 3 --gemm-detail gpurun_out/sd35_gemm_per_shape.txt > gpurun_out/sd35_bench.json 2> gpurun_out/sd35_bench.err; rc=$?
            echo "sd35 rc=$rc"; tail -c 1500 gpurun_out/sd35_bench.json; grep "host enqueue" gpurun_out/sd35_bench.err ;;
    lokr)   timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lokr_bench.json 2> gpurun_out/lokr_bench.err; rc=$?
            echo "lokr rc=$rc"; tail -c 1200 gpurun_out/lokr_bench.json ;;
    lora)   timeout -k 10 400 python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lora_bench.json 2> gpurun_out/lora_bench.err; rc=$?
            echo "lora rc=$rc"; tail -c 1200 gpurun_out/lora_bench.json ;;
    ddp)    # the data-parallel line rehearsed on one GPU (forced one-rank group): torch transport, native transport, sharded optimizer
            for mode in "YAT_COMM=torch" "YAT_COMM=native" "YAT_COMM=torch YAT_SHARD_OPTIMIZER=1"; do
              tag=$(echo "$mode" | tr ' =' '__')
              env YAT_DDP_FORCE=1 $mode timeout -k 10 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/ddp_$tag.json 2> gpurun_out/ddp_$tag.err; rc=$?
              echo "ddp [$mode] rc=$rc"; tail -c 2500 gpurun_out/ddp_$tag.json
              ok_to_continue $rc || break
            done ;;
    soak)   timeout -k 10 600 python bench.py --steps 1000 --warmup 10 --no-cpu-baseline --no-gemm-timer > gpurun_out/soak.json 2> gpurun_out/soak.err; rc=$?
            echo "soak rc=$rc"; tail -c 600 gpurun_out/soak.json ;;
    shards) timeout -k 10 600 python bench.py --data shards --steps 300 --warmup 10 --no-cpu-baseline > gpurun_out/shards.json 2> gpurun_out/shards.err; rc=$?
            echo "shards rc=$rc"; tail -c 600 gpurun_out/shards.json ;;
    ablation)   # WRONG results by construction -- only ms/step is read; two interleaved rounds on this box
      : > gpurun_out/step_ablation.txt
      for r in 1 2; do for a in none dwfwd dwbwd dwfwd,dwbwd la ln sdpa gate adamw dwfwd,dwbwd,la,ln,sdpa,gate; do
        ABLATE=$([ $a = none ] && echo "" || echo $a) timeout -k 10 200 python scripts/step_ablation.py --steps 16 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/abl.json 2> gpurun_out/abl.err; rc=$?
        echo "round $r  skipped: $(printf %-34s $a)  $(python3 -c "import json; print('%.2f' % json.loads(open('gpurun_out/abl.json').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null || echo fail) ms/step" | tee -a gpurun_out/step_ablation.txt
        ok_to_continue $rc || break 2
      done; done ;;
  esac
  ok_to_continue $rc || { echo "step $s killed by timeout (rc=$rc): stopping"; exit $rc; }
  [ "$s" = benchsmall ] && [ "$rc" -ne 0 ] && { echo "benchsmall failed: stopping"; exit $rc; }
done
exit 0
