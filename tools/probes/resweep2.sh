cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2   "; env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_RAMP=0" 20
run "SF_PIPE_RAMP=1" 20
run "SF_PIPE_STEAL=0.5" 20
run "SF_PIPE_STEAL=1" 20
run "SF_PIPE_FILL_STEAL=1" 20
run "SF_PIPE_PRE_STEAL=1" 20
run "SF_PIPE_PRE_STEAL=0,1" 20
run "SF_PIPE_STEAL=0.5" 100
run "SF_PIPE_RAMP=0" 100
