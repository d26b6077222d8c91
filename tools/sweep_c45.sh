# usage: tools/sweep_c45.sh "<cfg>:<split>:<group> ..."   (bench.py --config at 20 batches per setting)
cd $GRAFT_REPO_ROOT
for item in $1; do
  IFS=: read cfg sp g <<< "$item"
  echo -n "$cfg $sp G=$g   "
  SF_PIPE_CU_SPLIT=$sp SF_PIPE_GROUP=$g python bench.py --config $cfg --steps 20 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],2))"
done
