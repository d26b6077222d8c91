cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py --config $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_DRAIN_LAT=1" 20 C4
run "SF_PIPE_DRAIN_LAT=0" 20 C4
run "SF_PIPE_FILL=8" 20 C4
run "SF_PIPE_FILL=16" 20 C4
run "SF_PIPE_GROUP=7" 28 C4
run "SF_PIPE_GROUP=4" 28 C4
run "SF_PIPE_CU_SPLIT=rows3" 20 C4
run "SF_PIPE_CU_SPLIT=rows5" 20 C4
run "SF_PIPE_FILL=8" 20 C5
run "SF_PIPE_FILL=12" 20 C5
run "SF_PIPE_FILL=4" 20 C5
run "SF_PIPE_CU_SPLIT=rows2" 20 C5
