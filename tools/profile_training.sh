#!/bin/bash
# Run on the GPU box (via gpurun): the three training benches on the CURRENT kernels (HIP path + torch-eager reference formulation on the same
# GPU) and a rocprofv3 kernel-trace of each -> gpurun_out/<tag>_training*.{json,csv}.   Usage: tools/profile_training.sh <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
{
  echo -n '{"slotformer": '; timeout 600 python tools/bench_train.py --steps 20 --warmup 3 --eager 2>/dev/null | tail -1
  echo -n ', "slotformer_img": '; timeout 600 python tools/bench_train.py --steps 10 --warmup 2 --eager --img 2>/dev/null | tail -1
  echo -n ', "stosavi": '; timeout 600 python tools/bench_train_savi.py --steps 10 --warmup 2 --eager 2>/dev/null | tail -1
  echo -n ', "steve": '; timeout 900 python tools/bench_train_steve.py --steps 8 --warmup 2 --eager 2>/dev/null | tail -1
  echo '}'
} > $OUT/${TAG}_training.json
cat $OUT/${TAG}_training.json | head -c 1500; echo
cd /tmp && export TMPDIR=/tmp
for name in "" "_savi" "_steve"; do
  PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_train$name -o trace -- python $R/tools/bench_train$name.py --steps 5 --warmup 2 > $OUT/${TAG}_trace_train$name.log 2>&1
  f=$(find $OUT/${TAG}_trace_train$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${TAG}_training${name}_kernel_stats.csv && head -12 $f | cut -c1-150
  rm -rf $OUT/${TAG}_trace_train$name
done
