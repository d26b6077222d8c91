mkdir -p gpurun_out; rm -f gpurun_out/b1.log
for cfg in "rows4 12" "rows3 20" "rows2 20" "rows3 16" "rows3 12" "rows2 16"; do set -- $cfg
SF_PIPE_CU_SPLIT=$1 SF_PIPE_FILL=$2 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('split $1 fill $2 @20', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
cat gpurun_out/b1.log
