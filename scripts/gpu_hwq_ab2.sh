#!/bin/bash
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
run dry_default        YAT_DDP_FORCE=1 YAT_DDP_DRYRUN=1
run dry_prio_sides     YAT_DDP_FORCE=1 YAT_DDP_DRYRUN=1 YAT_STREAM_PRIORITY=-1
run dry_prio_all       YAT_DDP_FORCE=1 YAT_DDP_DRYRUN=1 YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1
run ddp_prio_sides     YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=-1
run ddp_prio_all       YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=-1 YAT_HP_MAIN=1
run plain              YAT_X=0
