cd $GRAFT_REPO_ROOT
run() { echo -n "$3 $1 steps $2   "; env $1 python bench.py $3 --steps $2 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],3))"; }
run "SF_BENCH_FORCE_DIST=1" 20 ""
run "SF_BENCH_FORCE_DIST=0" 20 ""
run "SF_BENCH_FORCE_DIST=1" 100 ""
run "SF_BENCH_FORCE_DIST=1" 20 "--config C4"
run "SF_BENCH_FORCE_DIST=1 SF_PIPE_FREE_SKIP=1" 20 "--config C4"
run "SF_BENCH_FORCE_DIST=1" 20 "--config C5 --batch 8"
run "SF_BENCH_FORCE_DIST=1 SF_PIPE_FREE_SKIP=1" 20 "--config C5 --batch 8"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('torchrun nproc 1', round(d['value']))"
