cd $GRAFT_REPO_ROOT
for k in 0 1 2 3 4 5; do echo -n "masked free streams, $k dummies  "; SF_PIPE_FREE_MASKED=1 SF_PIPE_FREE_DUMMY=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
