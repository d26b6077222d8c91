#!/usr/bin/env python
"""bench.py -- SAVi-encode + SlotFormer rollout throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic video resident in HBM:
StoSAVi encode of B x 6 frames (128x128, 7 slots, 2 Slot-Attention iterations) followed by a
50-step SlotFormer rollout (d=256, 4 layers, 8 heads) -- config C2 of SURVEY.md 8.
frames/s = n_gpus * B * (6 + 50) * steps / wall.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: videos shard on the batch axis, one process per GPU, weights replicated, NO
collective on the timed path -> weak scaling with B=32 per GPU.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak; bf16x3 issues 3 bf16 MFMA flops per algorithmic flop
PEAK_HBM_GBPS = 8000.0        # HBM3E spec peak (6.3 TB/s achievable)
T_BURN, T_ROLL, RES = 6, 50, 128
CLS_NAMES = ['conv_nhwc_implicit_gemm', 'conv_first', 'linear_gemm', 'slot_attn_iter', 'slot_update', 'attention', 'ffn_fused']


def c2_configs():
    import golden_util as gu
    return gu.C2_SAVI, gu.C2_ROLL


def build_models(dev):
    """Random-init weights of the C2 architecture (torch default initialisers, seed 0)."""
    import golden_util as gu
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    scfg, rcfg = c2_configs()
    torch.manual_seed(0)
    savi = build_model(gu.ParamsView(scfg)).eval()
    savi.testing = True
    roll = SlotRollouter(**rcfg['rollout_dict']).eval()
    return savi.to(dev), roll.to(dev)


def synthetic_img(B, seed=1234):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.rand(B, T_BURN, 3, RES, RES) * 2 - 1).astype(np.float32))


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json, newest round);
    None when no PMC summary has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))[kernel_key]['traffic_bytes_per_launch']
    except (KeyError, ValueError):
        return None


def read_profile(lib):
    out = {}
    for c, name in enumerate(CLS_NAMES):
        ms, n, w = C.c_double(), C.c_longlong(), C.c_double()
        lib.sf_profile_read(c, C.byref(ms), C.byref(n), C.byref(w))
        if n.value:
            out[name] = dict(launches=n.value, total_ms=ms.value, avg_us=1e3 * ms.value / n.value, work=w.value)
    return out


def cpu_baseline(sample_B):
    """The oracle (CPU port of the reference path, torch fp32) timed on the host cores on a
    bounded sample: sample_B videos of the same workload."""
    import golden_util as gu
    import oracle
    scfg, rcfg = c2_configs()
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(0)
    savi = build_model(gu.ParamsView(scfg))
    roll = SlotRollouter(**rcfg['rollout_dict'])
    ssd = {k: v.detach() for k, v in savi.state_dict().items()}
    rsd = {'rollouter.' + k: v.detach() for k, v in roll.state_dict().items()}
    img = synthetic_img(sample_B)
    noise = torch.randn(sample_B, T_BURN, 7, 128)

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
    return dict(value=sample_B * (T_BURN + T_ROLL) / dt, unit='frames/s', cores=best_t, kind='port',
                sample=f'oracle (torch-CPU fp32 restatement of the reference path), {sample_B} of 32 videos, '
                f'{reps} passes of encode 6 frames + 50-step rollout, {dt:.2f} s per pass, '
                f'{best_t} threads (fastest of 8/16/32/64) on a {ncpu}-hardware-thread host')


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
                    '(default: encode of batch i+1 overlaps the rollout graph of batch i on a second HIP stream)')
    ap.add_argument('--rollout-streams', type=int, default=1, help='batch groups rolled out concurrently on separate HIP streams')
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
    lib = _lib.lib()
    if args.precision:
        lib.sf_set_precision(1 if args.precision == 'bf16x3' else 0)
    prec = 'bf16x3' if lib.sf_get_precision() == 1 else 'f32'
    B = args.batch
    savi, roll = build_models(dev)
    img = synthetic_img(B, seed=1234 + rank).to(dev)
    N, D = 7, 128
    bufs = [torch.zeros(B, T_BURN + T_ROLL, N, D, device=dev) for _ in range(2)]
    buf = bufs[0]

    def encode(dst=None, feat_pre=None):
        noise = torch.randn(B, T_BURN, N, D, device=dev)  # fresh eps ~ N(0,1) per frame (savi.py:363-365), one launch
        post, _, _ = engine.savi_encode(savi, img, noise=noise, feat_pre=feat_pre)
        (buf if dst is None else dst)[:, :T_BURN].copy_(post)

    S = max(1, args.rollout_streams)
    assert B % S == 0
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []

    def rollout_eager():
        if S == 1:
            engine.rollout(roll, buf, T_BURN, T_ROLL)
            return
        cur = torch.cuda.current_stream()
        g = B // S
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                engine.rollout(roll, buf[i * g:(i + 1) * g], T_BURN, T_ROLL, ws_slot=i)
        for st in streams:
            cur.wait_stream(st)

    graph = None
    with torch.no_grad():
        log('first eager step')
        encode()
        torch.cuda.synchronize()
        log('encode ok')
        rollout_eager()
        torch.cuda.synchronize()
        log('rollout ok')
        if not args.no_graph:
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    rollout_eager()
                graph.replay()
                torch.cuda.synchronize()
                log('graph captured + replayed')
            except Exception as e:  # noqa: BLE001
                if rank == 0:
                    print(f'[bench] hipGraph capture failed ({e}); eager rollout', file=sys.stderr)
                graph = None

        def step():
            encode()
            if graph is not None:
                graph.replay()
            else:
                rollout_eager()

        # ---- software pipeline across batches: two streams, two slot buffers, one rollout graph per buffer ----
        overlap = (not args.no_overlap) and graph is not None and S == 1
        # rollout streams: SF_BENCH_ROLL_MASKS = comma list of hex mask words, one rollout stream per entry (each word
        # repeated over the 8 mask words); default = one stream on the complement of the encode mask
        cu_words = [int(w, 16) for w in os.environ.get('SF_BENCH_CU_SPLIT', 'ff').split(',')]
        cu_words = (cu_words * 8)[:8]
        cu_split = any(cu_words)
        roll_masks = [int(w, 16) for w in os.environ.get('SF_BENCH_ROLL_MASKS', '').split(',') if w]
        n_rs = max(1, len(roll_masks)) if cu_split else int(os.environ.get('SF_BENCH_ROLL_STREAMS', '1'))
        NB = n_rs + 1  # slot buffers / graphs / workspaces: one per rollout in flight + one being encoded
        if overlap:
            graphs = [graph]
            while len(bufs) < NB:
                bufs.append(torch.zeros_like(bufs[0]))
            try:
                for gi in range(1, NB):
                    engine.rollout(roll, bufs[gi], T_BURN, T_ROLL, ws_slot=gi)  # allocate its workspace before capture
                    torch.cuda.synchronize()
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1):
                        engine.rollout(roll, bufs[gi], T_BURN, T_ROLL, ws_slot=gi)  # own scratch: may run beside the others
                    graphs.append(g1)
            except Exception as e:  # noqa: BLE001
                log(f'graph capture {len(graphs)} failed ({e}); no overlap')
                overlap = False
        if overlap:
            prio = int(os.environ.get('SF_BENCH_ROLL_PRIO', '-1'))
            s_enc = torch.cuda.Stream(device=dev, priority=int(os.environ.get('SF_BENCH_ENC_PRIO', '0')))
            # CU partition (SF_BENCH_CU_SPLIT = hex word E, 0 = off): the encode stream gets the CUs whose bit is set
            # in E (repeated for each of the 8 mask words), the rollout stream(s) the complement.  Measured
            # (profiles/r01_probes.txt): byte j of every word behaves as shader engine j of every XCD; E = ff (one SE
            # = 64 CUs for the encode) beats the unpartitioned pipeline, partial bytes or per-word differences
            # unbalance the shader engines and lose 20-60 %.
            masked = []

            def masked_stream(ws):
                words = (C.c_uint * 8)(*[w & 0xffffffff for w in ws])
                h = C.c_void_p()
                _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), words, 8))
                masked.append(h)
                return torch.cuda.ExternalStream(h.value, device=dev)

            if cu_split:
                try:
                    s_enc = masked_stream(cu_words)
                    if roll_masks:
                        s_rolls = [masked_stream([w] * 8) for w in roll_masks]
                    elif os.environ.get('SF_BENCH_ROLL_UNMASKED', '0') == '1':  # experiment: rollout free to use every CU
                        s_rolls = [torch.cuda.Stream(device=dev, priority=prio)]
                    else:
                        s_rolls = [masked_stream([~w for w in cu_words])]
                except RuntimeError as e:  # CU masking unavailable: keep the pipeline, on shared CUs (and say so)
                    print(f'[bench] rank {rank}: CU-masked streams unavailable ({e}); pipelining on shared CUs', file=sys.stderr)
                    cu_split = False
            if not cu_split:
                n_rs = min(n_rs, NB - 1)
                s_rolls = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(n_rs)]

            # Work stealing (SF_BENCH_STEAL = number of time steps): the encode partition is the slower half, and the
            # rollout stream idles between its graph and the next encode's end -- so after the rollout of batch j it
            # computes the CNN features of the first `steal` time steps of batch j+2 on its own (larger) CU partition;
            # the encode of batch j+2 then skips those convolutions.  Two feature buffers.
            steal = int(os.environ.get('SF_BENCH_STEAL', '1')) if n_rs == 1 else 0
            steal = max(0, min(steal, T_BURN))
            feat_bufs = None
            if steal:
                feat_bufs = [engine.savi_cnn(savi, img, 0, steal, ws_slot=1) for _ in range(2)]
                torch.cuda.synchronize()

            def run_pipelined(n):
                cur = torch.cuda.current_stream()
                s_enc.wait_stream(cur)
                for sr in s_rolls:
                    sr.wait_stream(cur)
                ev_enc = [torch.cuda.Event() for _ in range(n)]
                ev_roll = [torch.cuda.Event(enable_timing=True) for _ in range(n)]   # also: completion time of every batch
                ev_pre = [torch.cuda.Event() for _ in range(n + 2)]
                # (the first two batches compute their own convolutions: stealing starts with batch 2, whose features are
                #  produced after the rollout of batch 0)
                for j in range(n):
                    if j == 0 and cu_split:
                        # pipeline fill: nothing else is running yet, so the first encode takes the whole chip (the
                        # calling stream) instead of the 64-CU partition; the masked encode stream starts after it
                        encode(bufs[0], None)
                        ev_enc[0].record(cur)
                        s_enc.wait_event(ev_enc[0])
                    else:
                        with torch.cuda.stream(s_enc):
                            if j >= NB:
                                s_enc.wait_event(ev_roll[j - NB])  # slot buffer j % NB is free once rollout j-NB is done
                            if steal and j >= 2:
                                s_enc.wait_event(ev_pre[j])
                            encode(bufs[j % NB], feat_bufs[j % 2] if (steal and j >= 2) else None)
                            ev_enc[j].record(s_enc)
                    s_roll = s_rolls[j % n_rs]
                    with torch.cuda.stream(s_roll):
                        s_roll.wait_event(ev_enc[j])
                        if j >= NB:
                            s_roll.wait_event(ev_roll[j - NB])  # same graph / buffer / workspace as batch j-NB
                        graphs[j % NB].replay()
                        ev_roll[j].record(s_roll)
                        if steal and j + 2 < n:
                            # feature buffer (j+2) % 2 == j % 2 was consumed by encode j, which this stream has waited for
                            engine.savi_cnn(savi, img, 0, steal, out=feat_bufs[j % 2], ws_slot=1)
                            ev_pre[j + 2].record(s_roll)
                cur.wait_stream(s_enc)
                for sr in s_rolls:
                    cur.wait_stream(sr)
                return ev_roll

        def barrier():
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        if overlap:
            run_pipelined(args.warmup)
        else:
            for _ in range(args.warmup):
                step()
        torch.cuda.synchronize()
        log('warmup done')
        # dominant kernel (conv implicit GEMM) + the HBM-bound SA iteration are event-timed live
        lib.sf_profile_enable((1 << 0) | (1 << 3))
        read_profile(lib)
        barrier()
        t0 = time.perf_counter()
        batch_done = None
        if overlap:
            batch_done = run_pipelined(args.steps)
        else:
            for _ in range(args.steps):
                step()
        barrier()
        elapsed = time.perf_counter() - t0
        log(f'timed region done: {elapsed:.3f}s')
        lib.sf_profile_enable(0)
        prof = read_profile(lib)

        # split timing (untimed extra): encode-only and rollout-only
        torch.cuda.synchronize()
        lib.sf_profile_enable((1 << 0) | (1 << 3))
        t1 = time.perf_counter()
        for _ in range(3):
            encode()
        torch.cuda.synchronize()
        t_enc = (time.perf_counter() - t1) / 3
        lib.sf_profile_enable(0)
        prof_iso = read_profile(lib)  # same kernels with nothing else on the GPU
        t1 = time.perf_counter()
        for _ in range(3):
            graph.replay() if graph is not None else rollout_eager()
        torch.cuda.synchronize()
        t_roll = (time.perf_counter() - t1) / 3
        part_ms = None
        if overlap and cu_split:
            # the two halves of the partitioned pipeline, each alone on its CU subset
            def timed_on(stream, fn, n=3):
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):
                    fn()
                    stream.synchronize()
                    t = time.perf_counter()
                    for _ in range(n):
                        fn()
                    stream.synchronize()
                return 1e3 * (time.perf_counter() - t) / n
            part_ms = {'encode_ms_on_its_cus': timed_on(s_enc, encode), 'rollout_ms_on_its_cus': timed_on(s_rolls[0], graph.replay)}
        pcie = None
        if args.pcie:
            # PCIe-inclusive variant (never `value`): frames start in pinned host memory and the slots end there
            img_h = img.cpu().pin_memory()
            out_h = torch.empty(bufs[0].shape, dtype=bufs[0].dtype).pin_memory()

            def step_pcie():
                img.copy_(img_h, non_blocking=True)
                step()
                out_h.copy_(buf, non_blocking=True)

            step_pcie()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step_pcie()
            torch.cuda.synchronize()
            t_p = (time.perf_counter() - t1) / args.steps
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            t_s = (time.perf_counter() - t1) / args.steps
            pcie = {'frames_per_s_host_to_host_serial': B * (T_BURN + T_ROLL) / t_p,
                    'frames_per_s_device_resident_serial': B * (T_BURN + T_ROLL) / t_s,
                    'h2d_bytes_per_batch': img.numel() * 4, 'd2h_bytes_per_batch': buf.numel() * 4,
                    'note': 'serial (no batch pipelining, copies on the compute stream): upper bound on the PCIe cost'}
        # the two kernels that make up a rollout layer, event-timed in one eager (un-graphed) rollout with nothing else
        # running -- inside the timed region they replay from a hipGraph, where HIP events cannot be inserted
        lib.sf_profile_enable((1 << 5) | (1 << 6))
        read_profile(lib)
        rollout_eager()
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        prof_roll = read_profile(lib)
        breakdown = None
        if args.breakdown:
            lib.sf_profile_enable(0x7f)
            encode()
            rollout_eager()
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
                'rollout_launch': 'hipGraph replay' if graph is not None else 'eager', 'rollout_streams': S,
                'pipelining': ('encode of batch i+1 (stream A) overlaps the rollout graph of batch i (stream B); every batch still '
                               'runs its full encode + 50-step rollout inside the timed region') if overlap else 'none',
                'work_stealing': (f'the CNN features of the first {steal} time step(s) of batch j+2 are computed on the rollout '
                                  'stream after the rollout of batch j') if (overlap and steal) else 'none',
                'cu_partition': (f'encode stream on CU mask {cu_words[0]:#x} x8 words ({8 * bin(cu_words[0]).count("1")} CUs), '
                                 + (f'{n_rs} rollout streams on masks ' + ','.join(f'{w:#x}' for w in roll_masks) if roll_masks
                                    else 'rollout stream on the complement')) if (overlap and cu_split) else 'none',
            },
            'encode_ms': 1e3 * t_enc,
            'rollout_ms': 1e3 * t_roll,
            'partitioned_ms': part_ms,
            'ms_per_step_distribution': step_dist,
            'encoded_frames_per_s': B * T_BURN / t_enc,
            'predicted_frames_per_s': B * T_ROLL / t_roll,
        }
        conv = prof.get('conv_nhwc_implicit_gemm')
        if conv:
            flops_per_launch = conv['work'] / conv['launches']
            ach = flops_per_launch / (conv['avg_us'] * 1e-6) / 1e12
            peak_chip = PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS
            # under the CU partition the encode stream owns a subset of the CUs during the timed region: the live
            # figure is priced against the MFMA peak of THOSE CUs; the isolated figure (kernel alone, default stream,
            # all 256 CUs) against the whole chip
            enc_cus = 8 * bin(cu_words[0]).count('1') if (overlap and cu_split) else 256
            peak = peak_chip * enc_cus / 256.0
            res['roofline'] = {
                'kernel': ('conv5x5_halo_kernel' if prec == 'bf16x3' else 'sf_gemm_kernel<128,64,...,conv_nhwc>') + ' (5x5 conv 64->64 @64x64, '
                + ('split-bf16 MFMA: 3 bf16 MFMA flops per algorithmic flop -> peak = 2500/3)' if prec == 'bf16x3' else 'exact f32 MFMA)'),
                'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': ach / peak, 'cus': enc_cus, 'peak_full_chip': peak_chip, 'traffic': pmc_traffic('conv_nhwc_implicit_gemm'),
                'traffic_unit': 'bytes/launch (rocprofv3 PMC 2*FETCH_SIZE+WRITE_SIZE, committed under profiles/)',
                'algorithmic_bytes_per_launch': 2 * 32 * 4096 * 64 * 4 + 64 * 1600 * 4,
                'flops_per_launch': flops_per_launch, 'avg_launch_us': conv['avg_us'], 'launches': conv['launches'],
                'avg_launch_us_isolated': prof_iso.get('conv_nhwc_implicit_gemm', {}).get('avg_us'),
                'frac_isolated': (flops_per_launch / (prof_iso['conv_nhwc_implicit_gemm']['avg_us'] * 1e-6) / 1e12 / peak_chip
                                  if 'conv_nhwc_implicit_gemm' in prof_iso else None),
                'note': 'achieved/frac are live over the timed region, where the encode runs on `cus` CUs beside the rollout graph '
                        'of the previous batch (peak = whole-chip peak x cus/256; the convolutions that work stealing moves to the '
                        'rollout stream are not part of this average); *_isolated is the same kernel alone on all 256 CUs '
                        '(vs peak_full_chip)',
            }
        # the rollout replays as ONE hipGraph (900 launches), so it is reported as a unit: algorithmic
        # FLOPs of SURVEY.md 8d (274.7 MFLOP per predicted frame per video, minus nothing: the last-layer
        # row pruning is an exact saving we do not credit) over the graph's wall time
        roll_flops = 274.7e6 * B * T_ROLL
        res['roofline_rollout_graph'] = {
            'kernel': 'hipGraph of the 50-step rollout: per step in-proj, 4 x (attention + out-proj partials, fused FFN), out-proj = 550 launches',
            'bound': 'latency (550 dependent launches, M = B*L = 1344 rows); MFMA roof shown for scale',
            'achieved': roll_flops / t_roll / 1e12, 'peak': (PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS),
            'unit': 'TFLOP/s', 'frac': roll_flops / t_roll / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS),
            'ms': 1e3 * t_roll, 'us_per_step': 1e6 * t_roll / T_ROLL,
        }
        rk = {}
        peak_bf3 = PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS
        for key, name in (('attention', 'attn_oproj_kernel (LN1 + q|k|v of one head + softmax(qk^T)v + out-proj partial; one WG per head x video)'),
                          ('ffn_fused', 'ffn_partial_kernel (sum of head partials + LN2 + FFN1 + ReLU + FFN2 chunk + last-arriver reduction)')):
            pk = prof_roll.get(key)
            if pk:
                fl = pk['work'] / pk['launches']
                tf = fl / (pk['avg_us'] * 1e-6) / 1e12
                rk[key] = {'kernel': name, 'bound': 'latency (M = 1344 rows; see DESIGN.md phase timings); MFMA roof for scale',
                           'achieved': tf, 'peak': peak_bf3, 'unit': 'TFLOP/s', 'frac': tf / peak_bf3, 'flops_per_launch': fl,
                           'avg_launch_us_isolated': pk['avg_us'], 'launches': pk['launches']}
        if rk:
            res['roofline_rollout_kernels'] = rk
        sa = prof.get('slot_attn_iter')
        if sa:
            bytes_per_launch = sa['work'] / sa['launches']
            gbps = bytes_per_launch / (sa['avg_us'] * 1e-6) / 1e9
            res['roofline_slot_attn'] = {
                'kernel': 'sa_attn_mfma_kernel<128> (one Slot-Attention iteration over K,V)', 'bound': 'hbm',
                'achieved': gbps, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': gbps / PEAK_HBM_GBPS,
                'traffic': pmc_traffic('slot_attn_iter'), 'bytes_per_launch': bytes_per_launch,
                'avg_launch_us_isolated': prof_iso.get('slot_attn_iter', {}).get('avg_us'), 'avg_launch_us': sa['avg_us'],
                'frac_isolated': (bytes_per_launch / (prof_iso['slot_attn_iter']['avg_us'] * 1e-6) / 1e9 / PEAK_HBM_GBPS
                                  if 'slot_attn_iter' in prof_iso else None),
                'cus': enc_cus if conv else 256,
                'note': 'live figure: the encode stream owns `cus` of the 256 CUs in the timed region; *_isolated: alone on the whole chip',
                'launches': sa['launches'],
            }
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
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
