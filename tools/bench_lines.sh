# the committed bench lines of a round besides the driver's command: tools/bench_lines.sh <tag>   -> gpurun_out/<tag>_bench*.json, <tag>_other_configs.txt
TAG=$1
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --config C4 --steps 20 --warmup 8 > gpurun_out/${TAG}_bench_C4.json 2>/dev/null
python bench.py --config C4 --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_C4_40steps.json 2>/dev/null
python bench.py --config C5 --steps 20 --warmup 8 > gpurun_out/${TAG}_bench_C5.json 2>/dev/null
python bench.py --config C5 --batch 8 --steps 20 --warmup 8 --no-cpu-baseline > gpurun_out/${TAG}_bench_C5_b8.json 2>/dev/null
python bench.py --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_C2_b4.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-dist > gpurun_out/${TAG}_bench_force_dist.json 2>/dev/null
python bench.py --pcie --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_pcie.json 2>/dev/null
for f in gpurun_out/${TAG}_bench_C*.json gpurun_out/${TAG}_bench_force_dist.json gpurun_out/${TAG}_bench_pcie.json; do python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1])
print('$f', round(d['value']), d['steps'], round(d['ms_per_step'],2), d['config'].get('rollout_units_of_the_timed_run'), d.get('roofline',{}).get('frac'), d.get('pcie_inclusive_frames_per_sec'), d.get('one_batch_latency_ms'))
"; done
python tools/bench_configs.py > gpurun_out/${TAG}_other_configs.txt 2>&1; tail -12 gpurun_out/${TAG}_other_configs.txt
