# bench sweeps of the token-stationary rollout inside the pipeline (tools only): bash tools/sweep_tok.sh
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-cpu-baseline --windows 3 $ARGS 2>gpurun_out/sw_$tag.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pm = d.get('partitioned_ms') or {}
print('$tag', 'steps', d['steps'], 'value', round(d['value']/1e3,1), 'k  ms/step', round(d['ms_per_step'],3), 'units', d['config'].get('rollout_units_of_the_timed_run'), 'unit_ms', round(pm.get('rollout_unit_ms_on_its_cus') or 0, 2), 'enc lane', pm.get('encode_lane_ms_on_its_cus'), 'E', d['config'].get('batches_per_encode'), flush=True)
" || tail -3 gpurun_out/sw_$tag.err; }
STEPS=20
run p2666 SF_PIPE_SIZES=2,6,6,6
run p2666_d2 SF_PIPE_SIZES=2,6,6,6 SF_PIPE_DRAIN_UNITS=2
run p4664 SF_PIPE_SIZES=4,6,6,4
run p26642 SF_PIPE_SIZES=2,6,6,4,2
run p6662_fill12 SF_PIPE_FILL=12
run p6662_h3 SF_PIPE_HYBRID=3
run p6662_ht0 SF_PIPE_HYBRID_TAIL=0
