cd $GRAFT_REPO_ROOT
for k in 0 1 2 3 4 5 6 7; do echo -n "skip=$k  "; SF_PIPE_PICK_STREAMS=0 SF_PIPE_FREE_SKIP=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
