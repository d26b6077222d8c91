cd $GRAFT_REPO_ROOT
for fp in 1 2 3 1 2 3; do echo -n "fill_par=$fp  "; SF_PIPE_FILL_PAR=$fp python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
