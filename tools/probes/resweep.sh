cd $GRAFT_REPO_ROOT
run() { echo -n "$1   "; env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_FILL=12"
run "SF_PIPE_FILL=4"
run "SF_PIPE_FILL=8"
run "SF_PIPE_FILL=16"
run "SF_PIPE_FILL=20"
run "SF_PIPE_FILL_PAR=3"
run "SF_PIPE_FILL_PAR=3 SF_PIPE_FILL=16"
run "SF_PIPE_CU_SPLIT=rows5"
run "SF_PIPE_CU_SPLIT=rows3"
run "SF_PIPE_DRAIN_LAT=1"
run "SF_PIPE_GROUP=5"
run "SF_PIPE_GROUP=2"
