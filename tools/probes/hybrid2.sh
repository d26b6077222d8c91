cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py --config $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
for c in C4 C5; do for k in 0 5 8; do run "SF_PIPE_HYBRID=$k" 20 $c; run "SF_PIPE_HYBRID=$k" 60 $c; done; done
run "SF_PIPE_HYBRID=5" 200 C2; run "SF_PIPE_HYBRID=0" 200 C2
