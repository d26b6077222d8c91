#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel-trace stats + PMC passes.
# Usage: tools/profile_round.sh <tag>     -> writes gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout 900 python bench.py --steps 40 --warmup 6 --breakdown --decode > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.log
tail -c 600 $OUT/${TAG}_bench.json | head -c 300; echo
cd /tmp && export TMPDIR=/tmp
# the default bench command (hipGraph rollout, CU-partitioned pipeline), minus the CPU leg, so that the traced kernel
# durations are the ones the bench line reports
BENCH="python $R/bench.py --steps 8 --warmup 4 --no-cpu-baseline --decode --decode-steps 2"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- $BENCH > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- $BENCH > $OUT/${TAG}_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -o pmc -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
python tools/summarize_profile.py $TAG
# keep what is committed (stats csv, summaries), drop the raw traces: gpurun merges at most 64 MiB back
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma
