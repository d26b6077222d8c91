#!/bin/bash
# Run on the GPU box: bench.py under several pipeline settings -> gpurun_out/sweep_<tag>.txt (one line per setting)
# usage: tools/sweep_bench.sh <tag> "<ENV1=.. ENV2=..>" "<...>" ...
TAG=$1; shift
OUT=gpurun_out/sweep_$TAG.txt
mkdir -p gpurun_out
: > $OUT
for cfg in "$@"; do
  line=$(env $cfg timeout 300 python bench.py --steps ${STEPS:-20} --warmup 6 --no-cpu-baseline ${BENCH_ARGS:-} 2> gpurun_out/sweep_err.txt | tail -1)
  python - "$cfg" "$line" >> $OUT <<'PY'
import json, sys
cfg, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    p = d.get('partitioned_ms') or {}
    print(f"{cfg:70s} {d['value']/1e3:8.1f} k frames/s  ms/step {d['ms_per_step']:.3f}  median {d['ms_per_step_distribution']['median']:.3f}  enc_lane {p.get('encode_ms_on_its_cus')}  roll_unit {p.get('rollout_unit_ms_on_its_cus'):.3f}  enc {d['encode_ms']:.3f} roll {d['rollout_ms']:.3f}")
except Exception as e:
    print(f"{cfg:70s} FAILED {e} {line[:200]}")
PY
done
cat $OUT
