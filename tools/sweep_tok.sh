# bench sweeps of the token-stationary rollout inside the pipeline (tools only): bash tools/sweep_tok.sh
run() { tag=$1; shift; env "$@" python bench.py --gpus 1 --steps $STEPS --warmup 5 --no-cpu-baseline --windows 3 $ARGS 2>gpurun_out/sw_$tag.err | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pm = d.get('partitioned_ms') or {}
print('$tag', 'steps', d['steps'], 'value', round(d['value']/1e3,1), 'k  ms/step', round(d['ms_per_step'],3), 'units', d['config'].get('rollout_units_of_the_timed_run'), 'unit_ms', round(pm.get('rollout_unit_ms_on_its_cus') or 0, 2), 'enc lane', pm.get('encode_lane_ms_on_its_cus'), 'E', d['config'].get('batches_per_encode'), flush=True)
" || tail -3 gpurun_out/sw_$tag.err; }
STEPS=20
ARGS="--config C4"
run c4_tok X=1
run c4_notok SF_PIPE_TOK=0
STEPS=40
run c4_tok40 X=1
run c4_notok40 SF_PIPE_TOK=0
ARGS="--config C5"
STEPS=20
run c5_tok X=1
run c5_notok SF_PIPE_TOK=0
ARGS="--config C5 --batch 8"
run c5b8_tok X=1
run c5b8_notok SF_PIPE_TOK=0
