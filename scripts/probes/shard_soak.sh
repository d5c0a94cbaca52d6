set -u
for mode in "YAT_COMM=torch YAT_SHARD_OPTIMIZER=1" "YAT_COMM=native YAT_SHARD_OPTIMIZER=1"; do
  tag=$(echo "$mode" | tr ' =' '__')
  env YAT_DDP_FORCE=1 $mode timeout -k 10 300 python bench.py --steps 500 --warmup 5 --no-cpu-baseline --no-gemm-timer --comm-steps 2 > gpurun_out/soak_$tag.json 2> gpurun_out/soak_$tag.err; rc=$?
  echo "soak [$mode] rc=$rc $(python3 -c "import json; d=json.loads(open('gpurun_out/soak_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['loss'], d['hbm_peak_gb'])")"
  [ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
timeout -k 10 300 python bench.py --lokr 8 --batch 32 --steps 6 --warmup 3 --no-cpu-baseline --lokr-pre-add > gpurun_out/lokr_pre_add.json 2> gpurun_out/lokr_pre_add.err; echo "lokr pre_add rc=$? $(python3 -c "import json; d=json.loads(open('gpurun_out/lokr_pre_add.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']); print(d['config']['workload'][:300])")"
