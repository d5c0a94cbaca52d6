#!/bin/bash
# Side benches of the other model families on one box: PixArt-Sigma (config 3), SD3.5-Medium (config 4), LoKr / LoRA B=32 (config 5).
set -u
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
WHAT="${1:-pixart sd35}"
for s in $WHAT; do
  case $s in
    pixart) timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 --gemm-detail gpurun_out/pixart_gemm_per_shape.txt > gpurun_out/pixart_bench.json 2> gpurun_out/pixart_bench.err; rc=$?
            echo "pixart rc=$rc"; tail -c 1500 gpurun_out/pixart_bench.json; grep "host enqueue" gpurun_out/pixart_bench.err ;;
    sd35)   timeout -k 10 500 python scripts/bench_sd35.py --steps 6 --warmup 3 --gemm-detail gpurun_out/sd35_gemm_per_shape.txt > gpurun_out/sd35_bench.json 2> gpurun_out/sd35_bench.err; rc=$?
            echo "sd35 rc=$rc"; tail -c 1500 gpurun_out/sd35_bench.json; grep "host enqueue" gpurun_out/sd35_bench.err ;;
    lokr)   timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lokr_bench.json 2> gpurun_out/lokr_bench.err; rc=$?
            echo "lokr rc=$rc"; tail -c 800 gpurun_out/lokr_bench.json ;;
    lora)   timeout -k 10 400 python bench.py --lora 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/lora_bench.json 2> gpurun_out/lora_bench.err; rc=$?
            echo "lora rc=$rc"; tail -c 800 gpurun_out/lora_bench.json ;;
    ddp)    # the data-parallel line rehearsed on one GPU (forced one-rank group), both transports
            YAT_DDP_FORCE=1 timeout -k 10 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/ddp_torch_bench.json 2> gpurun_out/ddp_torch_bench.err; rc=$?
            echo "ddp torch rc=$rc"; tail -c 2500 gpurun_out/ddp_torch_bench.json
            [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
            YAT_DDP_FORCE=1 YAT_COMM=native timeout -k 10 400 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-gemm-timer > gpurun_out/ddp_native_bench.json 2> gpurun_out/ddp_native_bench.err; rc=$?
            echo "ddp native rc=$rc"; tail -c 2500 gpurun_out/ddp_native_bench.json
            [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
            YAT_DDP_FORCE=1 timeout -k 10 400 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --no-gemm-timer > gpurun_out/ddp_lokr_bench.json 2> gpurun_out/ddp_lokr_bench.err; rc=$?
            echo "ddp lokr rc=$rc"; tail -c 2500 gpurun_out/ddp_lokr_bench.json ;;
  esac
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && { echo "step $s killed by timeout: stopping"; exit $rc; }
done
exit 0
