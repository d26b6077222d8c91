cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py --config $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
for g in 4 5 6 7; do run "SF_PIPE_GROUP=$g" 20 C4; run "SF_PIPE_GROUP=$g" 40 C4; done
run "SF_PIPE_GROUP=7 SF_PIPE_FILL=14" 20 C4
run "SF_PIPE_GROUP=3 SF_PIPE_FILL=12" 20 C2
run "SF_PIPE_GROUP=3 SF_PIPE_FILL=12" 100 C2
run "SF_PIPE_GROUP=6 SF_PIPE_FILL=12" 20 C2
run "SF_PIPE_GROUP=6 SF_PIPE_FILL=12" 100 C2
run "SF_PIPE_GROUP=4" 100 C2
