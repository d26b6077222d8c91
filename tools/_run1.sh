mkdir -p gpurun_out; rm -f gpurun_out/b1.log
for cfg in "0 20" "1 20" "2 20" "1 40" "2 40"; do set -- $cfg
SF_PIPE_ENC_SPLIT=$1 timeout 600 python bench.py --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('split $1 @$2', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
SF_PIPE_ENC_SPLIT=1 python tools/pipe_timeline.py 20 2>&1 | grep -v amdgpu >> gpurun_out/b1.log
cat gpurun_out/b1.log
