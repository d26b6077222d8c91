mkdir -p gpurun_out; rm -f gpurun_out/b1.log
for st in 0 0.5 1 1.5 2; do
SF_BENCH_STEAL=$st timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steal $st @20', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
for st in 0 1 2; do
SF_BENCH_STEAL=$st timeout 600 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steal $st @60', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
SF_BENCH_STEAL=1 timeout 300 python tools/pipe_timeline.py 20 2>&1 | grep -v amdgpu >> gpurun_out/b1.log
cat gpurun_out/b1.log
