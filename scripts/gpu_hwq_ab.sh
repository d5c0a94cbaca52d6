#!/bin/bash
# Hardware-queue aliasing A/B: the step with the data-parallel machinery forced on one rank, under
# different stream-priority / GPU_MAX_HW_QUEUES settings.  One line per configuration: ms/step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-timer > gpurun_out/hwq_$name.json 2> gpurun_out/hwq_$name.err
  python - "$name" <<'PY'
import json, sys
s = open(f"gpurun_out/hwq_{sys.argv[1]}.json").read()
try:
    d = json.loads(s[s.index('{"metric'):])
    print(f"{sys.argv[1]:28s} {d['ms_per_step']:8.2f} ms/step  {d['value']:7.2f} images/s")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run ddp_default            YAT_DDP_FORCE=1
run ddp_prio               YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1
run ddp_prio_q8            YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1 GPU_MAX_HW_QUEUES=8
run ddp_q16                YAT_DDP_FORCE=1 GPU_MAX_HW_QUEUES=16
run ddp_prio_sidesonly     YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=-1
run ddp_native_prio        YAT_DDP_FORCE=1 YAT_COMM=native YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1
run plain_default          YAT_X=0
run plain_prio             YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1
run plain_prio_q8          YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1 GPU_MAX_HW_QUEUES=8
