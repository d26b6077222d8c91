mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/t1.log
for cfg in "C2 -" "C5 -" "C5 rows2" "C4 -"; do set -- $cfg
  if [ "$2" = "-" ]; then unset SF_PIPE_CU_SPLIT; else export SF_PIPE_CU_SPLIT=$2; fi
  timeout 600 python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 split $2', round(d['value']), d['ms_per_step'], d['config']['rollout_opts'], d.get('partitioned_ms'))" >> gpurun_out/b1.log
done
unset SF_PIPE_CU_SPLIT
timeout 600 python bench.py --config C5 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C5 b8', round(d['value']), d['ms_per_step'], d['config']['rollout_opts'])" >> gpurun_out/b1.log
cat gpurun_out/t1.log gpurun_out/b1.log
