mkdir -p gpurun_out; rm -f gpurun_out/b1.log
for cfg in "4 8" "6 8" "6 12" "6 6" "5 10" "5 5"; do set -- $cfg
SF_BENCH_GROUP=$1 SF_PIPE_FILL=$2 SF_PIPE_ATTN_ROWS=128 SF_PIPE_FFN_TILE=1 timeout 600 python bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('C4 group $1 fill $2 tiles @20', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
cat gpurun_out/b1.log
