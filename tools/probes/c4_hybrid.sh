cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
for k in 5 4 3 2; do run "SF_PIPE_HYBRID=$k" 84 "--config C4"; done
run "SF_PIPE_HYBRID=3 SF_PIPE_FILL=14" 84 "--config C4"
run "SF_PIPE_HYBRID=5 SF_PIPE_FILL=14" 84 "--config C4"
python bench.py --config C5 --batch 8 --steps 48 --warmup 8 --no-cpu-baseline > gpurun_out/r03_bench_C5_b8_48steps.json 2>/dev/null
