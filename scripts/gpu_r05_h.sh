#!/bin/bash
# Round 5: torch transport without a stream of ours between the producer and the group's own stream: ddp tests + the forced
# one-rank rehearsal of both transports + plain, same box.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ddp_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/ddp_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/ddp_tests.log; [ $rc -ne 0 ] && exit $rc
for mode in plain torch native torch native; do
  case $mode in
    plain)  env= ;;
    torch)  env="YAT_DDP_FORCE=1 YAT_COMM=torch" ;;
    native) env="YAT_DDP_FORCE=1 YAT_COMM=native" ;;
  esac
  env $env timeout -k 10 400 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-gemm-timer > gpurun_out/fo_$mode.json 2> gpurun_out/fo_$mode.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  python3 - $mode <<'PY'
import json, sys
m = sys.argv[1]
d = json.loads(open(f"gpurun_out/fo_{m}.json").read().strip().splitlines()[-1])
c = d.get("comm") or {}
print(f"{m}: timed {d['ms_per_step']:.2f} ms/step | live {c.get('step_ms', float('nan')):.2f} collective off {c.get('step_ms_collective_off', float('nan')):.2f} "
      f"exposed {c.get('exposed_ms_per_step', float('nan')):.2f} comm stream {c.get('comm_stream_ms_per_step', float('nan')):.2f} ({c.get('transport', '-')})")
PY
done
