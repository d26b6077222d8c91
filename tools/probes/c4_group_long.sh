cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py --config $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_GROUP=4" 84 C4
run "SF_PIPE_GROUP=7" 42 C4
run "SF_PIPE_GROUP=7" 84 C4
run "SF_PIPE_GROUP=8" 80 C4
run "SF_PIPE_GROUP=10" 80 C4
run "SF_PIPE_GROUP=7 SF_PIPE_HYBRID=0" 84 C4
run "SF_PIPE_GROUP=3" 60 C5
run "SF_PIPE_GROUP=5" 60 C5
run "SF_PIPE_GROUP=4" 60 C5
