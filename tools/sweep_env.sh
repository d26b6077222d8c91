# usage: tools/sweep_env.sh "<bench args>" "ENV=.. ENV=.." "ENV=.." ...   (bench.py at 20 batches per environment)
cd $GRAFT_REPO_ROOT
args=$1; shift
for e in "$@"; do
  echo -n "$args | $e   "
  env $e python bench.py $args --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2))"
done
