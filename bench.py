#!/usr/bin/env python
"""bench.py -- SAVi-encode + SlotFormer rollout throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic video resident in HBM:
StoSAVi encode of B x 6 frames (128x128, 7 slots, 2 Slot-Attention iterations) followed by a
50-step SlotFormer rollout (d=256, 4 layers, 8 heads) -- config C2 of SURVEY.md 8.
frames/s = n_gpus * B * (6 + 50) * steps / wall.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

The timed schedule is the product one: `slotformer_amd.pipeline.EncodeRolloutPipeline` (the same object
`harness.extract_and_rollout` uses, tested bit-identical to the serial path in tests/test_pipeline_gpu.py), fed from a
ring of three DIFFERENT resident inputs, with the results of every batch copied out of the slot buffers.

Multi-GPU: videos shard on the batch axis, one process per GPU, weights replicated, NO
collective on the timed path -> weak scaling with B=32 per GPU.
"""
import argparse
import ctypes as C
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak; bf16x3 issues 3 bf16 MFMA flops per algorithmic flop
PEAK_HBM_GBPS = 8000.0        # HBM3E spec peak (6.3 TB/s achievable)
T_BURN, T_ROLL, RES = 6, 50, 128
N_SLOTS, SLOT_D = 7, 128
CLS_NAMES = ['conv_nhwc_implicit_gemm', 'conv_first', 'linear_gemm', 'slot_attn_iter', 'slot_update', 'attention', 'ffn_fused', 'seam']
ROLL_FLOPS_PER_FRAME = 274.7e6   # algorithmic FLOPs per predicted frame per video
ENC_FLOPS_PER_FRAME = 3.06e9     # SURVEY.md 8d / DESIGN.md 4: per encoded frame


def c2_configs():
    from slotformer_amd import configs
    return configs.C2_SAVI, configs.C2_ROLL


def build_models(dev):
    """Random-init weights of the C2 architecture (torch default initialisers, seed 0)."""
    from slotformer_amd import configs
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    scfg, rcfg = c2_configs()
    torch.manual_seed(0)
    savi = build_model(configs.ParamsView(scfg)).eval()
    savi.testing = True
    roll = SlotRollouter(**rcfg['rollout_dict']).eval()
    return savi.to(dev), roll.to(dev)


def synthetic_img(B, seed=1234):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.rand(B, T_BURN, 3, RES, RES) * 2 - 1).astype(np.float32))


def committed_profile(key):
    """Entry `key` of the newest committed PMC summary (profiles/r*_pmc_traffic.json); {} when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return {}
    try:
        d = json.load(open(files[-1])).get(key, {})
        d = dict(d)
        d['source'] = os.path.relpath(files[-1], ROOT)
        return d
    except ValueError:
        return {}


def dominant_kernel():
    """The kernel with the largest total device time in the newest committed rocprof summary of this command
    (profiles/r*_kernel_stats.csv) -> key of the roofline object that becomes `roofline`."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.csv')))
    names = {'conv5x5_halo': 'conv', 'ffn_partial_kernel': 'ffn_fused', 'ffn64_parts_kernel': 'ffn_fused', 'attn_oproj_kernel': 'attention',
             'seam_kernel': 'seam',
             'sa_attn_mfma_kernel': 'slot_attn'}
    if not files:
        return 'ffn_fused', None
    tot = {}
    try:
        for r in csv.DictReader(open(files[-1])):
            for pat, key in names.items():
                if pat in r['Name']:
                    tot[key] = tot.get(key, 0.0) + float(r['Percentage'])
    except (KeyError, ValueError):
        return 'ffn_fused', None
    if not tot:
        return 'ffn_fused', None
    key = max(tot, key=tot.get)
    return key, {'source': os.path.relpath(files[-1], ROOT), 'share_of_device_time_pct': {k: round(v, 2) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}}


def read_profile(lib):
    out = {}
    for c, name in enumerate(CLS_NAMES):
        ms, n, w = C.c_double(), C.c_longlong(), C.c_double()
        lib.sf_profile_read(c, C.byref(ms), C.byref(n), C.byref(w))
        if n.value:
            out[name] = dict(launches=n.value, total_ms=ms.value, avg_us=1e3 * ms.value / n.value, work=w.value)
    return out


def cpu_model_string():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def cpu_baseline(sample_B):
    """The oracle (CPU port of the reference path, torch fp32) timed on the host cores on a
    bounded sample: sample_B videos of the same workload."""
    import oracle
    from slotformer_amd import configs
    scfg, rcfg = c2_configs()
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(0)
    savi = build_model(configs.ParamsView(scfg))
    roll = SlotRollouter(**rcfg['rollout_dict'])
    ssd = {k: v.detach() for k, v in savi.state_dict().items()}
    rsd = {'rollouter.' + k: v.detach() for k, v in roll.state_dict().items()}
    img = synthetic_img(sample_B)
    noise = torch.randn(sample_B, T_BURN, N_SLOTS, SLOT_D)

    def run():
        with torch.no_grad():
            post = oracle.savi_encode(img, ssd, scfg, noise=noise)['post_slots']
            return oracle.rollouter_forward(post, T_ROLL, rsd, rcfg['rollout_dict'])

    # pick the thread count that is fastest on this host (more threads than ~32 only add
    # synchronisation cost at these sizes; 256 hardware threads ran >100x slower)
    ncpu = os.cpu_count() or 1
    best_t, best = None, None
    for th in [t for t in (8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        run()
        t0 = time.perf_counter()
        run()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, best_t = dt, th
    torch.set_num_threads(best_t)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 40):
        run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return dict(value=sample_B * (T_BURN + T_ROLL) / dt, unit='frames/s', cores=best_t, kind='port', cpu=cpu_model_string(),
                sample=f'oracle (torch-CPU fp32 restatement of the reference path), {sample_B} of 32 videos, '
                f'{reps} passes of encode 6 frames + 50-step rollout, {dt:.2f} s per pass, '
                f'{best_t} threads (fastest of 8/16/32/64) on a {ncpu}-hardware-thread host ({cpu_model_string()})')


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='videos per GPU')
    ap.add_argument('--no-graph', action='store_true', help='launch the rollout eagerly instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=8)
    ap.add_argument('--no-overlap', action='store_true', help='run encode and rollout of each batch back-to-back on one stream '
                    '(default: the batch pipeline of slotformer_amd/pipeline.py)')
    ap.add_argument('--precision', choices=['bf16x3', 'f32'], default=None, help='matrix arithmetic mode (default: library default = bf16x3)')
    ap.add_argument('--pcie', action='store_true', help='also time a host-to-host (PCIe-inclusive) variant; reported separately')
    ap.add_argument('--breakdown', action='store_true', help='extra untimed pass with every kernel class timed')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or os.environ.get('SF_BENCH_FORCE_DIST') == '1'  # the latter: exercise the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)  # RCCL

    from slotformer_amd import engine, _lib
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    lib = _lib.lib()
    if args.precision:
        lib.sf_set_precision(1 if args.precision == 'bf16x3' else 0)
    prec = 'bf16x3' if lib.sf_get_precision() == 1 else 'f32'
    B = args.batch
    savi, roll = build_models(dev)
    # a ring of three DIFFERENT resident inputs: consecutive batches never see the same frames
    ring = [synthetic_img(B, seed=1234 + 1000 * k + rank).to(dev) for k in range(3)]
    cu_word = os.environ.get('SF_BENCH_CU_SPLIT', 'ff')          # hex mask word, or rows<R> (pipeline.encode_mask_words)
    cu_word = cu_word if cu_word.startswith('rows') else int(cu_word, 16)
    steal = os.environ.get('SF_BENCH_STEAL')                   # None: the partition's default
    steal = None if steal is None else float(steal)
    partition = os.environ.get('SF_BENCH_PARTITION', 'pair')   # 'pair' | 'three' | 'two' | 'none' (pipeline.EncodeRolloutPipeline)

    with torch.no_grad():
        log('building the pipeline (first eager rollouts + graph capture)')
        pipe = EncodeRolloutPipeline(savi, roll, B, T_BURN, T_ROLL, encode_cu_word=cu_word, steal_steps=steal,
                                     use_graph=not args.no_graph, partition=partition)
        overlap = not args.no_overlap
        graph = pipe.graphs[0] if pipe.graphs else None

        def run(n, out):
            return pipe.run([ring[j % 3] for j in range(n)], None, out=out, serial=not overlap)

        def barrier():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        shape = (B, T_BURN + T_ROLL, N_SLOTS, SLOT_D)
        out_w = torch.empty((max(args.warmup, 1), ) + shape, device=dev)
        out_t = torch.empty((args.steps, ) + shape, device=dev)
        run(args.warmup, out_w)
        torch.cuda.synchronize()
        assert torch.isfinite(out_w[:args.warmup]).all(), 'non-finite slots in the warmup batches'
        log('warmup done')
        # conv + Slot-Attention launches are event-timed live (library brackets on the launch stream)
        # (every LIVE_EVERY-th launch of the two classes: two event records around a launch cost its stream a few microseconds,
        #  and the encode stream is the longer side of the pipeline)
        LIVE_EVERY = int(os.environ.get('SF_BENCH_LIVE_EVERY', '4'))
        lib.sf_profile_sample(LIVE_EVERY)
        lib.sf_profile_enable(int(os.environ.get('SF_BENCH_LIVE_MASK', str((1 << 0) | (1 << 3)))))
        read_profile(lib)
        barrier()
        t0 = time.perf_counter()
        run(args.steps, out_t)
        barrier()
        elapsed = time.perf_counter() - t0
        log(f'timed region done: {elapsed:.3f}s')
        lib.sf_profile_enable(0)
        lib.sf_profile_sample(1)
        prof = read_profile(lib)
        batch_done = pipe.completion_events if overlap else None
        assert torch.isfinite(out_t).all(), 'non-finite slots in the timed batches'
        # the three ring inputs give three different results (a stale slot buffer would repeat one)
        if args.steps >= 3:
            assert not torch.equal(out_t[0, :, :T_BURN], out_t[1, :, :T_BURN])

        # ---- untimed extras: the two halves alone, per-kernel event timings ----
        noise = torch.randn(B, T_BURN, N_SLOTS, SLOT_D, device=dev)

        def encode():
            post, _, _ = engine.savi_encode(savi, ring[0], noise=noise)
            pipe.bufs[0][:, :T_BURN].copy_(post)

        def rollout():
            graph.replay() if graph is not None else engine.rollout(roll, pipe.bufs[0], T_BURN, T_ROLL, ws_slot=('pipe', 0))

        def timed_on(stream, fn, n=3):
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                fn()
                stream.synchronize()
                t = time.perf_counter()
                for _ in range(n):
                    fn()
                stream.synchronize()
            return (time.perf_counter() - t) / n

        torch.cuda.synchronize()
        lib.sf_profile_enable((1 << 0) | (1 << 3))
        t_enc = timed_on(torch.cuda.current_stream(), encode)
        lib.sf_profile_enable(0)
        prof_iso = read_profile(lib)  # same kernels with nothing else on the GPU
        t_roll = timed_on(torch.cuda.current_stream(), rollout)
        part_ms = None
        if overlap and pipe.cu_split:
            def lane_encode(li):
                st, lo, hi = pipe.lanes[li]
                return lambda: pipe._encode(ring[0], noise, pipe.bufs[0], None, lo, hi, li)

            part_ms = {'encode_lane_ms_on_its_cus': [round(1e3 * timed_on(pipe.lanes[li][0], lane_encode(li)), 4) for li in range(len(pipe.lanes))],
                       'encode_lane_videos': [hi - lo for _, lo, hi in pipe.lanes],
                       'rollout_ms_on_its_cus': 1e3 * timed_on(pipe.s_roll, rollout)}
            part_ms['encode_ms_on_its_cus'] = max(part_ms['encode_lane_ms_on_its_cus'])
        # the rollout layer kernels, event-timed: (a) alone on the whole chip in one eager rollout, (b) LIVE in a pipelined
        # pass with the product schedule (encode stream busy on its CUs) but eager launches -- inside the timed region they
        # replay from a hipGraph, where HIP events cannot be inserted between the kernels
        lib.sf_profile_enable((1 << 5) | (1 << 6) | (1 << 7))
        read_profile(lib)
        engine.rollout(roll, pipe.bufs[0], T_BURN, T_ROLL, ws_slot=('pipe', 0))
        torch.cuda.synchronize()
        prof_roll = read_profile(lib)
        prof_roll_live = {}
        if overlap:
            graphs, pipe.graphs = pipe.graphs, []
            pipe.run([ring[j % 3] for j in range(4)], None, out=out_w if out_w.shape[0] >= 4 else None)
            torch.cuda.synchronize()
            pipe.graphs = graphs
            prof_roll_live = read_profile(lib)
        lib.sf_profile_enable(0)
        pcie = None
        if args.pcie:
            # PCIe-inclusive variant (never `value`): frames start in pinned host memory and the slots end there
            img_h = [r.cpu().pin_memory() for r in ring]
            out_h = torch.empty(shape, dtype=torch.float32).pin_memory()
            stage = torch.empty_like(ring[0])

            def step_pcie(j):
                stage.copy_(img_h[j % 3], non_blocking=True)
                post, _, _ = engine.savi_encode(savi, stage, noise=noise)
                pipe.bufs[0][:, :T_BURN].copy_(post)
                rollout()
                out_h.copy_(pipe.bufs[0], non_blocking=True)

            step_pcie(0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for j in range(args.steps):
                step_pcie(j)
            torch.cuda.synchronize()
            t_p = (time.perf_counter() - t1) / args.steps
            pcie = {'frames_per_s_host_to_host_serial': B * (T_BURN + T_ROLL) / t_p,
                    'frames_per_s_device_resident_serial': B * (T_BURN + T_ROLL) / (t_enc + t_roll),
                    'h2d_bytes_per_batch': ring[0].numel() * 4, 'd2h_bytes_per_batch': out_h.numel() * 4,
                    'note': 'serial (no batch pipelining, copies on the compute stream): upper bound on the PCIe cost'}
        breakdown = None
        if args.breakdown:
            lib.sf_profile_enable(0xff)
            encode()
            engine.rollout(roll, pipe.bufs[0], T_BURN, T_ROLL, ws_slot=('pipe', 0))
            torch.cuda.synchronize()
            lib.sf_profile_enable(0)
            breakdown = read_profile(lib)

    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    if rank == 0:
        frames = world * B * (T_BURN + T_ROLL) * args.steps
        step_dist = None
        if batch_done is not None and len(batch_done) > 2:
            # device time between the completions of consecutive batches inside the timed region (rank 0)
            gaps = sorted(batch_done[j - 1].elapsed_time(batch_done[j]) for j in range(1, len(batch_done)))
            pick = lambda q: gaps[min(len(gaps) - 1, int(q * len(gaps)))]  # noqa: E731
            step_dist = {'p10': pick(0.10), 'median': pick(0.50), 'p90': pick(0.90), 'n': len(gaps)}
        peak_chip = PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS
        peak_note = ('split-bf16 MFMA: 3 bf16 MFMA flops per algorithmic flop -> roof = 2500/3 TFLOP/s' if prec == 'bf16x3'
                     else 'exact f32 MFMA')
        nl = len(roll.transformer_encoder.layers)
        launches_per_graph = 1 + T_ROLL * 2 * nl - (T_ROLL - 1)   # the seam launch carries two kernels' worth
        step_flops = B * (T_BURN * ENC_FLOPS_PER_FRAME + T_ROLL * ROLL_FLOPS_PER_FRAME)
        res = {
            'metric': 'rollout frames/sec at B=32, 128x128, 7 slots, 6+50 steps (SAVi-encode + SlotFormer rollout)',
            'value': frames / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 storage; matmul/conv on split-bf16 MFMA (bf16x3: hi*hi+hi*lo+lo*hi, f32 accumulate)' if prec == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'C2: CLEVRER StoSAVi 128x128 (7 slots, D=128, 2 SA iters, stochastic kernels, MLP '
                'predictor) encode of 6 burn-in frames + SlotFormer (d=256, 4 layers, 8 heads, ffn 1024, L=42) '
                '50-step rollout; random-init weights',
                'batch_per_gpu': B, 'global_batch': B * world, 'burn_in': T_BURN, 'rollout': T_ROLL,
                'parallelism': f'dp{world}: videos sharded on the batch axis, no collective on the timed path',
                'inputs': 'ring of 3 different resident batches; the slots of every batch are copied out of the slot buffers',
                'schedule': 'slotformer_amd.pipeline.EncodeRolloutPipeline (product code, tests/test_pipeline_gpu.py)',
                'rollout_launch': f'hipGraph replay ({launches_per_graph} kernel nodes)' if graph is not None else 'eager',
                'pipelining': ('encode of batch i+1 (stream A) overlaps the rollout graph of batch i (stream B); every batch still '
                               'runs its full encode + 50-step rollout inside the timed region') if overlap else 'none',
                'work_stealing': (f'the CNN features of the first {pipe.steal:g} time step(s) (average) of batch j+{2 * len(pipe.roll_streams)} are '
                                  'computed on the rollout stream of batch j right after its rollout') if (overlap and pipe.steal) else 'none',
                'cu_partition': ('two rollout streams (batches j and j+1 roll out side by side) on CU rows 0-4 of all four shader engines of every '
                                 'XCD (160 CUs), the encode stream on rows 5-7 (96 CUs)' if pipe.partition == 'pair' else
                                 ('rollout stream: CU rows 0-6 of shader engines 1-3 of every XCD (168 CUs); encode lane 0: shader engine 0 '
                                  f'(64 CUs, {pipe.lanes[0][2] - pipe.lanes[0][1]} videos of every batch); encode lane 1: CU row 7 of shader engines 1-3 '
                                  f'(24 CUs, {pipe.lanes[-1][2] - pipe.lanes[-1][1]} videos)') if pipe.partition == 'three' else
                                 (f'encode stream on CU mask {cu_word if isinstance(cu_word, str) else hex(cu_word)} ({pipe.encode_cus} CUs, the same '
                                  'number in every XCD), rollout stream on the complement')) if (overlap and pipe.cu_split) else 'none',
            },
            'encode_ms': 1e3 * t_enc,
            'rollout_ms': 1e3 * t_roll,
            'partitioned_ms': part_ms,
            'ms_per_step_distribution': step_dist,
            'encoded_frames_per_s': B * T_BURN / t_enc,
            'predicted_frames_per_s': B * T_ROLL / t_roll,
            'whole_step_tflops': step_flops * args.steps * world / elapsed / 1e12,
            'whole_step_flops': step_flops,
        }
        # ---- roofline objects, one per hot kernel; `roofline` = the one that dominates the committed rocprof summary of this
        #      command (profiles/r*_kernel_stats.csv), the others stay as secondary keys ----
        objs = {}
        for key, name in (('ffn_fused', 'ffn_partial_kernel / ffn64_parts_kernel (sum of the 4 head-pair partials + LN2 + FFN1 + ReLU + FFN2 on a 32- or '
                           '64-row tile x 256-wide hidden chunk; chunk partials out, last-arriver reduction on the last layer only)'),
                          ('attention', 'attn_oproj_kernel (LN1 + q|k|v of a head pair + softmax(qk^T)v + out-proj partial; one workgroup per '
                           '(head pair, video))'),
                          ('seam', 'seam_kernel (last-layer FFN + step boundary of step s and the layer-0 attention of step s+1 in one grid, '
                           'tile-local write-through hand-off)')):
            iso, live = prof_roll.get(key), prof_roll_live.get(key)
            if not iso:
                continue
            fl = iso['work'] / iso['launches']
            src = live or iso
            cus = pipe.rollout_cus if (live and pipe.cu_split) else 256
            tf = fl / (src['avg_us'] * 1e-6) / 1e12
            pm = committed_profile(key)
            objs[key] = {
                'kernel': name, 'bound': 'mfma', 'achieved': tf, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': tf / peak_chip,
                'peak_note': peak_note + '; whole-chip roof (the launch has 128-168 workgroups, one per CU)',
                'flops_per_launch': fl, 'avg_launch_us': src['avg_us'], 'launches': src['launches'], 'cus_available': cus,
                'measured': ('HIP events around every launch on the rollout stream in a pipelined pass with the timed schedule '
                             '(encode stream busy on its CUs), eager launches' if live else 'HIP events around every launch, kernel alone'),
                'avg_launch_us_isolated': iso['avg_us'], 'frac_isolated': fl / (iso['avg_us'] * 1e-6) / 1e12 / peak_chip,
                'traffic': pm.get('traffic_bytes_per_launch'),
                'mfma_busy_frac': pm.get('mfma_busy_frac'), 'avg_launch_us_rocprof': pm.get('avg_launch_us_trace'), 'pmc_source': pm.get('source'),
                'limiter': 'per-CU ingest (~100 GB/s per workgroup: weight fragments + activations re-fetched by every workgroup) and '
                           'dependent-launch latency, not the matrix pipe (DESIGN.md 4)',
            }
        roll_flops = ROLL_FLOPS_PER_FRAME * B * T_ROLL
        launches_per_graph = 1 + T_ROLL * 2 * nl - (T_ROLL - 1 if 'seam' in prof_roll else 0)
        res['roofline_rollout_graph'] = {
            'kernel': f'hipGraph of the 50-step rollout: ring init + per step {nl} x (attention + out-proj partials, fused FFN), the last FFN '
                      f'(+ step boundary) sharing a launch with the next step\'s first attention = {launches_per_graph} launches',
            'bound': f'latency ({launches_per_graph} dependent launches, M = B*L = {B * 42} rows); MFMA roof shown for scale',
            'achieved': roll_flops / t_roll / 1e12, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': roll_flops / t_roll / 1e12 / peak_chip,
            'ms': 1e3 * t_roll, 'us_per_step': 1e6 * t_roll / T_ROLL, 'seam_timeouts': int(lib.sf_seam_timeouts()),
        }
        conv = prof.get('conv_nhwc_implicit_gemm')
        if conv:
            flops_live = conv['work'] / conv['launches']        # mean over the lanes' launches (each a share of the batch)
            ach = flops_live / (conv['avg_us'] * 1e-6) / 1e12
            # CUs one live launch runs on: the lanes work side by side, each on its own CUs
            enc_cus = pipe.encode_cus / len(pipe.lanes) if (overlap and pipe.cu_split) else 256
            iso = prof_iso.get('conv_nhwc_implicit_gemm')
            flops_per_launch = iso['work'] / iso['launches'] if iso else flops_live
            ach_iso = flops_per_launch / (iso['avg_us'] * 1e-6) / 1e12 if iso else None
            pm = committed_profile('conv_nhwc_implicit_gemm')
            objs['conv'] = {
                'kernel': ('conv5x5_halo_kernel' if prec == 'bf16x3' else 'sf_gemm_kernel<128,64,...,conv_nhwc>') + ' (5x5 conv 64->64 @64x64; ' + peak_note + ')',
                'bound': 'lds-read / mfma (matrix pipes busy ~40 % of the launch at 2.4 GHz, 55 % at the 1.86 GHz the chip sustains in this '
                         'kernel; fragment reads from LDS and the un-overlapped halo fill bound it, DESIGN.md 4 and 7)',
                'achieved': ach_iso, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': (ach_iso / peak_chip) if ach_iso else None,
                'avg_launch_us': iso['avg_us'] if iso else None,
                'measured': 'HIP events around the launches (library brackets on the launch stream): `achieved` = the kernel alone on the whole '
                            'chip in the untimed pass of this run, every launch; `live` = inside the timed region on the encode partition, '
                            f'every {LIVE_EVERY}th launch (the brackets cost the stream they sit on a few microseconds each)',
                'live': {'achieved': ach, 'cus': enc_cus, 'avg_launch_us': conv['avg_us'], 'launches': conv['launches'],
                         'flops_per_launch': flops_live, 'frac_of_partition_peak': ach / (peak_chip * enc_cus / 256.0),
                         'note': 'inside the timed region, beside the rollout graph of the previous batch; `cus` = CUs per launch (the encode '
                                 'lanes run side by side on their own CUs, each on its share of the videos: mean over the lanes); stolen '
                                 'convolutions (rollout stream) are not part of this average'},
                'traffic': pm.get('traffic_bytes_per_launch'), 'mfma_busy_frac': pm.get('mfma_busy_frac'),
                'avg_launch_us_rocprof': pm.get('avg_launch_us_trace'), 'pmc_source': pm.get('source'),
                'traffic_unit': 'bytes/launch (rocprofv3 PMC 2*FETCH_SIZE+WRITE_SIZE)',
                'algorithmic_bytes_per_launch': 2 * 32 * 4096 * 64 * 4 + 64 * 1600 * 4, 'flops_per_launch': flops_per_launch,
            }
        sa = prof.get('slot_attn_iter')
        if sa:
            gbps = sa['work'] / sa['launches'] / (sa['avg_us'] * 1e-6) / 1e9
            iso = prof_iso.get('slot_attn_iter')
            bytes_per_launch = iso['work'] / iso['launches'] if iso else sa['work'] / sa['launches']
            objs['slot_attn'] = {
                'kernel': 'sa_attn_mfma_kernel<128> (one Slot-Attention iteration over K,V)', 'bound': 'hbm',
                'achieved': bytes_per_launch / (iso['avg_us'] * 1e-6) / 1e9 if iso else None, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                'frac': (bytes_per_launch / (iso['avg_us'] * 1e-6) / 1e9 / PEAK_HBM_GBPS) if iso else None,
                'avg_launch_us': iso['avg_us'] if iso else None,
                'traffic': committed_profile('slot_attn_iter').get('traffic_bytes_per_launch'), 'bytes_per_launch': bytes_per_launch,
                'live': {'achieved': gbps, 'avg_launch_us': sa['avg_us'], 'launches': sa['launches'],
                         'bytes_per_launch': sa['work'] / sa['launches'],
                         'cus': pipe.encode_cus / len(pipe.lanes) if (overlap and pipe.cu_split) else 256},
            }
        dom, dom_info = dominant_kernel()
        if dom not in objs:
            dom = 'ffn_fused' if 'ffn_fused' in objs else (next(iter(objs)) if objs else None)
        if dom:
            res['roofline'] = dict(objs.pop(dom), dominant_by=dom_info)
        for k, v in objs.items():
            res['roofline_' + k] = v
        if breakdown:
            res['kernel_breakdown_one_step'] = breakdown
        if pcie:
            res['pcie_inclusive'] = pcie
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline ...')
            res['cpu_baseline'] = cpu_baseline(args.cpu_sample)
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a pipe): push it out first so
        # that the JSON line is the last thing on stdout
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)
    pipe.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
