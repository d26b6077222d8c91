# the committed bench lines of a round: tools/bench_lines.sh <tag>   -> gpurun_out/<tag>_bench*.json
TAG=$1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.log
python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_20steps.json 2>/dev/null
python bench.py --steps 100 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_100steps.json 2>/dev/null
python bench.py --config C4 --steps 20 --warmup 8 > gpurun_out/${TAG}_bench_C4.json 2>/dev/null
python bench.py --config C5 --steps 20 --warmup 8 > gpurun_out/${TAG}_bench_C5.json 2>/dev/null
python bench.py --config C4 --steps 42 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_C4_42steps.json 2>/dev/null
python bench.py --config C5 --batch 8 --steps 48 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5_b8_48steps.json 2>/dev/null
python bench.py --config C5 --batch 8 --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5_b8.json 2>/dev/null
python bench.py --pcie --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_pcie.json 2>/dev/null
for f in gpurun_out/${TAG}_bench_*.json; do python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value']), d['steps'], round(d['ms_per_step'],2), d.get('roofline',{}).get('frac'), d.get('cpu_baseline',{}) and d['cpu_baseline'].get('value'), d.get('pcie_inclusive_frames_per_sec'))
"; done
