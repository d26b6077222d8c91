#!/bin/bash
# kernel trace of a 60-batch pipelined run (tools/pipe_timeline.py) -> per-queue duration / gap summary (tools/unit_gap_probe.py)   usage: tools/gap_trace.sh <tag> [ENV=..]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
env "$@" timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_qt -o trace -- python $R/tools/pipe_timeline.py 60 > $R/gpurun_out/${TAG}_timeline.txt 2>&1
cd $R
python tools/unit_gap_probe.py gpurun_out/${TAG}_qt 150 40 > gpurun_out/${TAG}_gaps.txt 2>&1
rm -rf gpurun_out/${TAG}_qt
tail -12 gpurun_out/${TAG}_timeline.txt; cat gpurun_out/${TAG}_gaps.txt
