cd $GRAFT_REPO_ROOT
for o in A,B,L D,A,B,L A,D,B,L A,B,D,L A,B,L,D L,A,B A,L,B A,B,L,D,D D,D,A,B,L; do for k in 0 1; do echo -n "order $o skip $k   "; SF_PIPE_QORDER=$o SF_PIPE_FREE_SKIP=$k python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done; done
