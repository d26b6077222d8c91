cd $GRAFT_REPO_ROOT
for c in C2 C2 C4 C5; do echo -n "$c default (skip 1 + collision check)  "; python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; done
python -m pytest tests/test_pipeline_gpu.py tests/test_harness_gpu.py -q -m gpu -x 2>&1 | tail -2
