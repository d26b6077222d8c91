mkdir -p gpurun_out; rm -f gpurun_out/*.log
for cfg in "128 0 4" "128 1 4" "128 2 4" "128 0 6" "128 1 6" "0 1 4" "128 0 3"; do set -- $cfg
  SF_PIPE_ATTN_ROWS=$1 SF_BENCH_STEAL=$2 SF_BENCH_GROUP=$3 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('attn_rows $1 steal $2 group $3', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
SF_PIPE_ATTN_ROWS=128 timeout 300 python tools/pipe_timeline.py 20 > gpurun_out/tl.log 2>&1
SF_PIPE_ATTN_ROWS=128 SF_BENCH_STEAL=1 timeout 300 python tools/pipe_timeline.py 20 > gpurun_out/tl2.log 2>&1
cat gpurun_out/b1.log gpurun_out/tl.log gpurun_out/tl2.log
