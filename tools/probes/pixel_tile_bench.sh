cd $GRAFT_REPO_ROOT
for t in 128 64 128 64; do echo -n "tile=$t  "; SF_PIXEL_TILE=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
for t in 128 64; do echo -n "C5 tile=$t  "; SF_PIXEL_TILE=$t python bench.py --config C5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
