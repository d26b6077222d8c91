#!/bin/bash
# rocprofv3 PMC pass over any command (counters alone: no trace domains) -> gpurun_out/<tag>_pmc.txt   usage: tools/pmc_cmd.sh <tag> "<counters>" <kernel-substring> <command...>
TAG=$1; CNT=$2; KSUB=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PYTHONPATH=$R timeout 600 rocprofv3 --pmc $CNT --output-format csv -d $R/gpurun_out/${TAG}_pmc -o pmc -- "$@" > $R/gpurun_out/${TAG}_pmc.log 2>&1
cd $R
python - $TAG "$KSUB" <<'PY'
import csv, glob, sys, collections
tag, ksub = sys.argv[1], sys.argv[2]
f = glob.glob(f'gpurun_out/{tag}_pmc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if ksub in r['Kernel_Name']:
        a = acc[r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
out = [f"{k:32s} avg {v[0] / max(v[1], 1):16.1f}  n={v[1]}" for k, v in sorted(acc.items())]
open(f'gpurun_out/{tag}_pmc.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $R/gpurun_out/${TAG}_pmc
