mkdir -p gpurun_out; rm -f gpurun_out/b1.log gpurun_out/t1.log gpurun_out/p1.log
timeout 900 python -m pytest tests/test_rollout_opts_gpu.py tests/test_engine_gpu.py -x -q -m gpu -k "row_tile or throughput or roll_ or rollout" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > gpurun_out/t1.log
SF_LF_DBG=16 timeout 300 python tools/attn_rows_probe.py 128 50 2>&1 | grep -v amdgpu > gpurun_out/p1.log
for cfg in "C2 1" "C2 2" "C5 1" "C5 2" "C4 1" "C4 2"; do set -- $cfg
SF_PIPE_FFN_TILE=$2 timeout 600 python bench.py --config $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 ffn_tile $2 @20', round(d['value']), d['ms_per_step'], d['partitioned_ms']['rollout_unit_ms_on_its_cus'])" >> gpurun_out/b1.log
done
cat gpurun_out/t1.log gpurun_out/p1.log gpurun_out/b1.log
