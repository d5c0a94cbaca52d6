#!/bin/bash
# Hardware-queue aliasing A/B: resident and trainer-fed step, with and without the data-parallel
# machinery forced on one rank (YAT_DDP_DRYRUN=1: hooks, events and streams without the collective itself), default stream
# policy against YAT_STREAM_PRIORITY=0 (everything at the normal level) / -1 (compute streams at the high level).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-timer $EXTRA > gpurun_out/hwq_$name.json 2> gpurun_out/hwq_$name.err
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
run plain              YAT_X=0
run ddp                YAT_DDP_FORCE=1
run ddp_normal_level   YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=0
run ddp_dry            YAT_DDP_FORCE=1 YAT_DDP_DRYRUN=1
run ddp_dry_normal     YAT_DDP_FORCE=1 YAT_DDP_DRYRUN=1 YAT_STREAM_PRIORITY=0
EXTRA="--data shards"
run shards             YAT_X=0
run shards_high_level  YAT_STREAM_PRIORITY=-1
run shards_ddp         YAT_DDP_FORCE=1
run shards_ddp_normal  YAT_DDP_FORCE=1 YAT_STREAM_PRIORITY=0
