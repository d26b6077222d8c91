mkdir -p gpurun_out; rm -f gpurun_out/b1.log gpurun_out/t1.log
timeout 1500 python -m pytest tests/test_pipeline_gpu.py tests/test_harness_gpu.py tests/test_dist_gpu.py tests/test_slot_io_gpu.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t1.log
for cfg in "C2 20" "C2 40" "C4 20" "C5 20"; do set -- $cfg
timeout 600 python bench.py --config $1 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 @$2', round(d['value']), d['ms_per_step'])" >> gpurun_out/b1.log
done
cat gpurun_out/t1.log gpurun_out/b1.log
