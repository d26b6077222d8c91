mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t1.log
for cfg in "0 0" "128 0" "128 1" "0 1"; do set -- $cfg
  SF_PIPE_ATTN_ROWS=$1 SF_BENCH_STEAL=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('attn_rows $1 steal $2', round(d['value']), d['ms_per_step'], d.get('partitioned_ms'))" >> gpurun_out/b1.log
done
SF_PIPE_ATTN_ROWS=128 timeout 300 python tools/pipe_timeline.py 20 > gpurun_out/tl.log 2>&1
cat gpurun_out/t1.log gpurun_out/b1.log gpurun_out/tl.log
