#!/bin/bash
# Round 5: gemm256 epilogue by lane exchange (v_permlane16_swap) instead of the LDS slab round trip: bit identity against the
# slab build, kernel tests, per-shape times (SANA + PixArt shapes), step A/B, PixArt / SD3.5 benches with both.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
V=yat_amd/build/variants
ok() { [ "$1" -ne 124 ] && [ "$1" -ne 137 ]; }
timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_a.txt 2> gpurun_out/hash_a.err; rc=$?; ok $rc || exit $rc
YAT_HIP_LIB=$V/libyat_slab.so timeout -k 10 300 python scripts/gemm_hash.py > gpurun_out/hash_b.txt 2> gpurun_out/hash_b.err; rc=$?; ok $rc || exit $rc
if diff -q gpurun_out/hash_a.txt gpurun_out/hash_b.txt > /dev/null; then echo "HASH IDENTICAL ($(wc -l < gpurun_out/hash_a.txt) lines, $(grep -c WRONG gpurun_out/hash_a.txt) wrong)"; else echo "HASH DIFFERS"; diff gpurun_out/hash_a.txt gpurun_out/hash_b.txt | head -10; fi
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm or linear or wgrad" > gpurun_out/gemm_tests.log 2>&1; rc=$?
tail -n 3 gpurun_out/gemm_tests.log; ok $rc || exit $rc
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/probe_epi.txt
export PROBE_CUSTOM="nn:8192:11200:2240,nn:8192:6720:2240,nn:8192:2240:2240,nt:8192:5600:2240,nt:8192:2240:11200,tt:6720:2240:8192,nn:32768:4608:1152,nn:32768:3456:1152,nn:32768:1152:4608,nt:32768:1152:1152,nn:35432:6144:1536,nn:35432:1536:6144"
for lib in product slab product slab; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  echo "== $lib" >> gpurun_out/probe_epi.txt
  timeout -k 10 400 python scripts/gemm_sustained_probe.py >> gpurun_out/probe_epi.txt 2> gpurun_out/probe_epi.err; rc=$?
  ok $rc || exit $rc
done
unset YAT_HIP_LIB PROBE_CUSTOM
cut -c1-120 gpurun_out/probe_epi.txt
bash scripts/gpu_ab.sh "YAT_X=0" "YAT_HIP_LIB=$V/libyat_slab.so" 30 || exit $?
bash scripts/gpu_ab.sh "YAT_HIP_LIB=$V/libyat_slab.so" "YAT_X=0" 30 || exit $?
for lib in product slab; do
  if [ $lib = product ]; then unset YAT_HIP_LIB; else export YAT_HIP_LIB=$V/libyat_$lib.so; fi
  timeout -k 10 400 python scripts/bench_pixart.py --steps 8 --warmup 3 > gpurun_out/pixart_$lib.json 2> gpurun_out/pixart_$lib.err; rc=$?; ok $rc || exit $rc
  timeout -k 10 500 python scripts/bench_sd35.py --steps 6 --warmup 3 > gpurun_out/sd35_$lib.json 2> gpurun_out/sd35_$lib.err; rc=$?; ok $rc || exit $rc
  python3 -c "
import json
for m in ('pixart','sd35'):
    d=json.loads(open('gpurun_out/%s_$lib.json' % m).read().strip().splitlines()[-1]); print('$lib', m, round(d['ms_per_step'],2), 'ms/step', round(d['value'],2), 'img/s')"
done
