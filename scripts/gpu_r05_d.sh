#!/bin/bash
# Round 5 checkpoint: the whole GPU suite on the pruned tree, the side benches, kernel stats of configs 3 / 4 / 5.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/gpu_round.sh "tests smoke" || exit $?
bash scripts/gpu_models.sh "pixart sd35 lokr lora ddp" || exit $?
bash scripts/gpu_prof_models.sh "pixart sd35" > gpurun_out/prof_models.txt 2>&1 || exit $?
bash scripts/gpu_prof_lokr.sh
