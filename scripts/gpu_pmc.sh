#!/bin/bash
# PMC passes (separate runs, counters only + kernel trace): SQ wave/wait/LDS/MFMA counters, then HBM FETCH_SIZE, then WRITE_SIZE.
#   gpu_pmc.sh ["sq fetch write"] [SCRIPT ARGS...]      default target: the serialized bench (the launch mix of its roofline pass);
#   another target, e.g. a micro-benchmark: gpu_pmc.sh "sq fetch write" scripts/dwconv_bench.py
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
# streams serialized, one forward chain: the launch mix of bench.py's roofline pass (which the per-launch traffic is quoted for)
export YAT_SERIAL=1
PASSES="${1:-sq fetch write}"
[ $# -gt 0 ] && shift
ARGS="${*:-bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-timer}"
run() { # name counters...
  name=$1; shift
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/pmc_$name -o pmc -- python3 $ARGS > gpurun_out/pmc_$name.json 2> gpurun_out/pmc_$name.err
  rc=$?; echo "pmc $name rc=$rc"; [ $rc -ne 124 ] && [ $rc -ne 137 ]
}
case " $PASSES " in *" sq "*) ;; *) SKIP_SQ=1;; esac
[ -z "${SKIP_SQ:-}" ] && { run sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1; }
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE || exit 1
find gpurun_out -name "*counter_collection*" | head
