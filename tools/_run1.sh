mkdir -p gpurun_out; rm -f gpurun_out/*.log
timeout 900 python -m pytest tests/test_rollout_opts_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t1.log
SF_LF_DBG=16 timeout 300 python tools/attn_rows_probe.py 128 50 > gpurun_out/p1.log 2>&1
for cfg in "C2 0 0" "C2 128 1" "C5 128 0" "C5 128 1" "C4 128 1"; do set -- $cfg
  SF_PIPE_ATTN_ROWS=$2 SF_PIPE_FFN_TILE=$3 timeout 600 python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 attn_rows $2 ffn_tile $3', round(d['value']), d['ms_per_step'], d.get('partitioned_ms'))" >> gpurun_out/b1.log
done
cat gpurun_out/t1.log gpurun_out/p1.log gpurun_out/b1.log
