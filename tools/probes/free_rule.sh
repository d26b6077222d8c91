cd $GRAFT_REPO_ROOT
for v in "SF_PIPE_FREE_SKIP=0" "SF_PIPE_FREE_SKIP=1"; do
  echo "== $v"
  env SF_PIPE_PICK_STREAMS=0 $v python tools/two_pipes_probe.py 2>&1 | tail -5 | head -3
  env SF_PIPE_PICK_STREAMS=0 $v python tools/pcie_probe.py 20 2>&1 | grep -E "call [0-3]|device-resident"
  env SF_PIPE_PICK_STREAMS=0 $v python bench.py --pcie --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bench --pcie', round(d['value']), [round(x) for x in d['pcie_inclusive']['frames_per_s_host_to_host_calls_3_to_6']])"
  env SF_PIPE_PICK_STREAMS=0 $v python bench.py --steps 100 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('100 steps', round(d['value']))"
done
