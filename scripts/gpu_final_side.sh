#!/bin/bash
# Side benches + soak on the final tree (each step under its own timeout; stop after a kill).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 500 "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err; rc=$?; echo "$name rc=$rc $(python3 -c "import json,sys; d=json.loads(open('gpurun_out/$name.json').read().strip().splitlines()[-1]); print('%.2f ms/step %.1f img/s loss %.4f' % (d['ms_per_step'], d['value'], d['loss']))" 2>/dev/null)"; [ $rc -ne 124 ] && [ $rc -ne 137 ] || exit $rc; }
run soak_resident python bench.py --steps 400 --warmup 20 --no-cpu-baseline
run soak_shards python bench.py --data shards --steps 200 --warmup 10 --no-cpu-baseline
run pixart python scripts/bench_pixart.py --steps 8 --warmup 3 --roofline-steps 1
run sd35 python scripts/bench_sd35.py --steps 6 --warmup 3 --roofline-steps 1
run lokr_b32 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline
run lora_b32 python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline
