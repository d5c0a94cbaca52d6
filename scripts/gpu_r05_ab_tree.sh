#!/bin/bash
# Round 5: same-box A/B of the default bench between this tree and the tree of an earlier commit built under
# yat_amd/build/variants/old_tree (git worktree): does the gemm256 change for the second operand pair cost the main path anything?
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ROOT=$PWD
for r in 1 2 3; do
  for t in new old; do
    d=$ROOT; [ $t = old ] && d=$ROOT/yat_amd/build/variants/old_tree
    ( cd $d && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/ab_tree_${t}_$r.json 2> $ROOT/gpurun_out/ab_tree_${t}_$r.err ); rc=$?
    [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
    python3 -c "
import json; d=json.loads(open('$ROOT/gpurun_out/ab_tree_${t}_$r.json').read().strip().splitlines()[-1]); print('$t $r', round(d['ms_per_step'],2), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1), d['loss'])"
  done
done
