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
# the driver's bench command itself (--steps 20 --warmup 5: hipGraph rollout units, CU-partitioned pipeline; no decode stage), minus the CPU leg and
# with three timed windows, so that the traced kernel durations and device-time shares are those of the headline's timed region
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --windows 3 --min-timed-s 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o trace -- $BENCH > $OUT/${TAG}_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o pmc -- $BENCH > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o pmc -- $BENCH > $OUT/${TAG}_pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -o pmc -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1
# SQ counter pass (counters alone) of the same command: issue / wait / LDS / matrix-pipe cycles of the hot kernels, with the source tree of THIS build
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_pmc_sq -o pmc -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
python tools/summarize_profile.py $TAG
python - $TAG <<'PY' > $OUT/${TAG}_sq_counters.txt
import csv, glob, sys, collections
sys.path.insert(0, '.')
import bench
tag = sys.argv[1]
fs = glob.glob(f'gpurun_out/{tag}_pmc_sq/**/*counter_collection.csv', recursive=True)
print('# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES (one pass, counters only)')
print(f'# of python bench.py --steps 20 --warmup 5 --no-cpu-baseline --windows 3 --min-timed-s 0 [tools/profile_round.sh]; per-dispatch averages over the named kernel\'s launches (source tree {bench.source_tree_hash()}).')
print('# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs, SQ_LDS_* LDS-array cycles summed over CUs (MI355X_MICROARCH.md).')
if fs:
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    KS = ('layer_tok_kernel', 'ffn_qkv_tile', 'ffn_partial_kernel', 'attn_core', 'qkv_rows', 'conv5x5_ws', 'conv5x5_rows4', 'sa_attn_tile', 'sa_attn_planes', 'sa_slot_update_mfma', 'pixel_feat_stream', 'pixel_feat_tok', 'conv_first')
    for r in csv.DictReader(open(fs[0])):
        for k in KS:
            if k in r['Kernel_Name']:
                a = acc[k][r['Counter_Name']]
                a[0] += float(r['Counter_Value']); a[1] += 1
    for k in KS:
        if k in acc:
            print('==', k)
            for c, v in sorted(acc[k].items()):
                print(f'{c:32s} avg {v[0] / max(v[1], 1):16.1f}  n={v[1]}')
PY
# keep what is committed (stats csv, summaries), drop the raw traces: gpurun merges at most 64 MiB back
cp $(find $OUT/${TAG}_trace -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_sq
