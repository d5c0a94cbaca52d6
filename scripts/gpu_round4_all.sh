#!/bin/bash
# round-4 state: full GPU tests, smoke, headline bench (with cpu_baseline), configs 3 / 4 / 5 + LoRA
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/tests_gpu.log 2>&1; rc=$?
echo "exit=$rc" >> gpurun_out/tests_gpu.log; grep -E "passed|failed" gpurun_out/tests_gpu.log | tail -2
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?; tail -3 gpurun_out/smoke.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; echo "bench rc=$rc"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 > gpurun_out/pixart_bench.json 2> gpurun_out/pixart_bench.err; rc=$?
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python scripts/bench_sd35.py --steps 6 --warmup 2 > gpurun_out/sd35_bench.json 2> gpurun_out/sd35_bench.err; rc=$?
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lokr_bench.json 2> gpurun_out/lokr_bench.err; rc=$?
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 400 python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lora_bench.json 2> gpurun_out/lora_bench.err; rc=$?
python - <<'PY'
import json
for name in ("bench", "pixart_bench", "sd35_bench", "lokr_bench", "lora_bench"):
    try:
        d = json.load(open(f"gpurun_out/{name}.json")); r = d.get("roofline", {})
        print("%-13s %.2f ms/step  %.2f img/s  mfma_util_step %.3f  gemm %.1f ms %.0f TF/s  host %.1f ms%s" % (
            name, d["ms_per_step"], d["value"], d.get("mfma_util_step", 0), r.get("gemm_ms_per_step_serialized", 0), r.get("achieved", 0),
            d.get("host_enqueue_ms_per_step", 0), ("  cpu %.3f img/s on %d cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])) if d.get("cpu_baseline") else ""))
    except Exception as e:
        print(name, "failed:", e)
PY
