cd $GRAFT_REPO_ROOT
for g in 4 7 6 5 8; do
  for st in 28; do
  echo -n "C4 G=$g steps=$st   "
  SF_PIPE_GROUP=$g python bench.py --config C4 --steps $st --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2))"
  done
done
