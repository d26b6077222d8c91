cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2   "; env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_HYBRID=5" 20
run "SF_PIPE_HYBRID=5 SF_PIPE_HYBRID_TAIL=3" 20
run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=3" 20
run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=2" 20
run "SF_PIPE_HYBRID=3 SF_PIPE_HYBRID_TAIL=3" 20
run "SF_PIPE_HYBRID=3 SF_PIPE_HYBRID_TAIL=3 SF_PIPE_FILL=8" 20
run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=3 SF_PIPE_FILL=8" 20
run "SF_PIPE_HYBRID=5 SF_PIPE_HYBRID_TAIL=3" 100
run "SF_PIPE_HYBRID=5 SF_PIPE_FILL=8" 100
run "SF_PIPE_HYBRID=5 SF_PIPE_FILL=16" 100
