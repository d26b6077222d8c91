mkdir -p gpurun_out
for cfg in "C2 40 r03_bench" "C2 20 r03_bench_20steps" "C2 100 r03_bench_100steps" "C4 20 r03_bench_C4" "C5 20 r03_bench_C5"; do set -- $cfg
  timeout 900 python bench.py --config $1 --steps $2 --warmup 5 > gpurun_out/$3.json 2>/dev/null
done
timeout 900 python bench.py --config C5 --batch 8 --steps 20 --warmup 5 > gpurun_out/r03_bench_C5_b8.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --pcie > gpurun_out/r03_bench_pcie.json 2>/dev/null
bash tools/trace_cmd.sh r03_C5 python $GRAFT_REPO_ROOT/bench.py --config C5 --steps 8 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
bash tools/trace_cmd.sh r03_C4 python $GRAFT_REPO_ROOT/bench.py --config C4 --steps 8 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
mkdir -p gpurun_out/keep; for c in C4 C5; do f=$(find gpurun_out/r03_${c}_qt -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/keep/r03_kernel_stats_${c}.csv; done; rm -rf gpurun_out/r03_C4_qt gpurun_out/r03_C5_qt
python tools/pipe_timeline.py 20 2>&1 | grep -v amdgpu > gpurun_out/r03_timeline.txt
timeout 600 python tools/bench_configs.py 2>&1 | grep -v amdgpu > gpurun_out/r03_other_configs.txt
SF_LF_DBG=16 timeout 300 python tools/attn_rows_probe.py 128 50 2>&1 | grep -v amdgpu > gpurun_out/r03_rows_probe.txt
for f in r03_bench r03_bench_20steps r03_bench_100steps r03_bench_C4 r03_bench_C5 r03_bench_C5_b8 r03_bench_pcie; do python -c "import json,sys; d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['roofline']['kernel'][:40], (d.get('pcie_inclusive') or {}).get('frames_per_s_host_to_host'))"; done
cat gpurun_out/r03_timeline.txt
