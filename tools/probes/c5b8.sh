cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3), d['config']['rollout_launch'][:60])"; }
run "SF_BENCH_GROUP=8" 120 "--config C5 --batch 8"
run "SF_BENCH_GROUP=10" 120 "--config C5 --batch 8"
run "SF_BENCH_GROUP=12" 120 "--config C5 --batch 8"
run "SF_BENCH_GROUP=7" 126 "--config C4"
run "SF_BENCH_GROUP=9" 126 "--config C4"
run "SF_BENCH_GROUP=11" 121 "--config C4"
