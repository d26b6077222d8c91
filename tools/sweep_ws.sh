# bench sweeps with the weights-stationary convolution on the masked encode lane (tools only)
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-cpu-baseline --windows 3 2>gpurun_out/sw_$tag.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pm = d.get('partitioned_ms') or {}
print('$tag', 'steps', d['steps'], 'value', round(d['value']/1e3,1), 'k  ms/step', round(d['ms_per_step'],3), 'units', d['config'].get('rollout_units_of_the_timed_run'), 'unit_ms', round(pm.get('rollout_unit_ms_on_its_cus') or 0, 2), 'enc lane', pm.get('encode_lane_ms_on_its_cus'), flush=True)
" || tail -3 gpurun_out/sw_$tag.err; }
STEPS=60
run base
run g5 SF_PIPE_GROUP=5
run g5_h3 SF_PIPE_GROUP=5 SF_PIPE_HYBRID=3
run g5_h6 SF_PIPE_GROUP=5 SF_PIPE_HYBRID=6
run g7 SF_PIPE_GROUP=7
run h5 SF_PIPE_HYBRID=5
run h6 SF_PIPE_HYBRID=6
run h8 SF_PIPE_HYBRID=8
STEPS=20
run base
run g5 SF_PIPE_GROUP=5
run h6 SF_PIPE_HYBRID=6
run fill12 SF_PIPE_FILL=12
run fill24 SF_PIPE_FILL=24
