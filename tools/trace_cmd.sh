#!/bin/bash
# rocprofv3 kernel-trace stats of any command -> gpurun_out/<tag>_stats.txt   usage: tools/trace_cmd.sh <tag> <command...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_qt -o trace -- "$@" > $R/gpurun_out/${TAG}_qt.log 2>&1
cd $R
python - $TAG <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/{tag}_qt/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = []
for r in rows[:24]:
    out.append(f"{r['Name'].split('(')[0][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} pct {float(r['Percentage']):6.2f}")
open(f'gpurun_out/{tag}_stats.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
