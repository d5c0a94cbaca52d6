#!/bin/bash
# forced one-rank data-parallel rehearsal: torch transport vs the native transport (gloo rendezvous group), plus plain step
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); c = d.get("comm") or {}
print("%s: timed %.2f ms/step | live %.2f  collective off %.2f  exposed %.2f  comm stream %.2f  (%s)" % (
    sys.argv[1].split("/")[-1], d["ms_per_step"], c.get("step_ms", float("nan")), c.get("step_ms_collective_off", float("nan")),
    c.get("exposed_ms_per_step", float("nan")), c.get("comm_stream_ms_per_step", float("nan")), c.get("transport", c.get("error", "-"))))
PY
}
for rep in 1 2; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-gemm-timer > gpurun_out/nt_plain_$rep.json 2> gpurun_out/nt_plain_$rep.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  show gpurun_out/nt_plain_$rep.json
  YAT_DDP_FORCE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-gemm-timer > gpurun_out/nt_torch_$rep.json 2> gpurun_out/nt_torch_$rep.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  show gpurun_out/nt_torch_$rep.json
  YAT_DDP_FORCE=1 YAT_COMM=native timeout -k 10 300 python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-gemm-timer > gpurun_out/nt_native_$rep.json 2> gpurun_out/nt_native_$rep.err; rc=$?
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
  show gpurun_out/nt_native_$rep.json
done
# LoKr after the small-wgrad staging fix
timeout -k 10 600 python -m pytest tests/test_lokr_gpu.py tests/test_lora_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/tests_lokr.log 2>&1; rc=$?; echo "lokr tests rc=$rc"; tail -2 gpurun_out/tests_lokr.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer > gpurun_out/lokr_bench2.json 2> gpurun_out/lokr_bench2.err; rc=$?
python -c "
import json; d=json.load(open('gpurun_out/lokr_bench2.json')); print('lokr: %.1f ms/step %.1f img/s host %.1f ms' % (d['ms_per_step'], d['value'], d['host_enqueue_ms_per_step']))"
