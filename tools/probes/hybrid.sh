cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2   "; env $1 python bench.py --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_PIPE_HYBRID=0" 20; run "SF_PIPE_HYBRID=0" 40; run "SF_PIPE_HYBRID=0" 100
for k in 4 5 6; do run "SF_PIPE_HYBRID=$k" 20; run "SF_PIPE_HYBRID=$k" 40; run "SF_PIPE_HYBRID=$k" 100; done
run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=8" 20; run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=8" 40; run "SF_PIPE_HYBRID=4 SF_PIPE_HYBRID_TAIL=8" 100
