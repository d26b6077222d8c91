#!/bin/bash
# one bench.py line reduced to its value: tools/bv.sh <bench.py arguments>   (A/B runs on one box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
python $R/bench.py --no-cpu-baseline "$@" 2>/tmp/bv.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
w=d.get('ms_per_step_windows',{})
print('$*', '->', round(d['value']/1e3,1), 'k frames/s', round(d['ms_per_step'],3), 'ms  p10/p90', round(w.get('p10',0),3), round(w.get('p90',0),3), ' lane', d.get('partitioned_ms',{}).get('encode_lane_ms_on_its_cus'), 'unit', round(d.get('partitioned_ms',{}).get('rollout_unit_ms_on_its_cus',0),2))
" || tail -5 /tmp/bv.err
