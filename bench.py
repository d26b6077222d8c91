#!/usr/bin/env python
"""bench.py -- SAVi-encode + SlotFormer rollout throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic video resident in HBM: slot extraction of the burn-in
frames (StoSAVi / STEVE encode) followed by the autoregressive SlotFormer rollout.  Default = config C2 of SURVEY.md 8
(the configuration BASELINE.json's metric is quoted on): B = 32 videos of 128x128, 7 slots, 6 burn-in frames + 50 rollout
steps, d = 256, 4 layers.      frames/s = n_gpus * B * (burn_in + rollout) * steps / wall.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config C4        # Physion STEVE + 8-layer SlotFormer, B = 16, 6 + 40
    python bench.py --config C5        # PHYRE SAVi + SingleStepSlotRollouter, B = 64 (--batch 8: the per-GPU share on 8 GPUs), 1 + 80

The timed schedule is the product one: `slotformer_amd.pipeline.EncodeRolloutPipeline` (the same object
`harness.extract_and_rollout` uses, tested bit-identical to the serial path in tests/test_pipeline_gpu.py), fed from a
ring of three DIFFERENT resident inputs, with the results of every batch copied out of the slot buffers.

Multi-GPU: videos shard on the batch axis, one process per GPU, weights replicated, NO collective on the timed path
 -> weak scaling with B videos per GPU (`--batch 4 --gpus 8` is the strong-scaling split of one global batch
of 32; the pipeline is not tuned for B = 4 and no 8-GPU node was available to measure either curve).

Roofline objects: `frac` of every kernel object divides its algorithmic work per launch by the GRAPH-REPLAY launch
duration of the committed rocprof trace of this command (profiles/r*_pmc_traffic.json: avg_launch_us_trace; the timed
region replays hipGraphs, where HIP events cannot be placed between kernels) -- recomputable from the committed CSVs by
division.  The durations measured live in this run with HIP events around eager launches are reported beside it
(`avg_launch_us_events`; they carry 3-5 us of event overhead per launch).
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
CLS_NAMES = ['conv_nhwc_implicit_gemm', 'conv_first', 'linear_gemm', 'slot_attn_iter', 'slot_update', 'attention', 'ffn_fused', 'seam', 'deconv_head', 'layer_tok']


def bench_configs():
    """name -> (savi cfg, rollout cfg, single-step?, default B, burn-in, rollout steps, resolution, description)"""
    from slotformer_amd import configs as c
    steve = dict(c.C4_STEVE)
    # the image side (dVAE, token decoder) is not on this path: the reference's Physion sizes, never run here
    steve.update(dvae_dict=dict(down_factor=4, vocab_size=4096, dvae_ckp_path=''),
                 dec_dict=dict(dec_type='slate', dec_num_layers=4, dec_num_heads=4, dec_d_model=192),
                 loss_dict=dict(use_img_recon_loss=False))
    return {
        'C2': (c.C2_SAVI, c.C2_ROLL, False, 32, 6, 50, 128,
               'C2: CLEVRER StoSAVi 128x128 (7 slots, D=128, 2 SA iters, stochastic kernels, MLP predictor) encode of 6 burn-in '
               'frames + SlotFormer (d=256, 4 layers, 8 heads, ffn 1024, L=42) 50-step rollout'),
        'C4': (steve, c.C4_ROLL, False, 16, 6, 40, 128,
               'C4: Physion STEVE 128x128 (6 slots, D=192, 2 SA iters, Transformer+LSTM predictor) encode of 6 burn-in frames + '
               'SlotFormer (d=256, 8 layers, 8 heads, ffn 1024, L=36) 40-step rollout'),
        'C5': (c.C5_SAVI, c.C5_ROLL, True, 64, 1, 80, 128,
               'C5: PHYRE SAVi 128x128 (8 slots, D=128, Transformer+LSTM predictor, deterministic kernels) encode of 1 frame + '
               'SingleStepSlotRollouter (d=256, 8 layers, window growing to 6 frames = 48 tokens) 80-step rollout'),
    }


def build_models(dev, cfg):
    """Random-init weights of the configuration's architecture (torch default initialisers, seed 0)."""
    from slotformer_amd import configs
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter, SingleStepSlotRollouter
    scfg, rcfg, single = cfg[0], cfg[1], cfg[2]
    torch.manual_seed(0)
    savi = build_model(configs.ParamsView(scfg)).eval()
    savi.testing = True
    roll = (SingleStepSlotRollouter if single else SlotRollouter)(**rcfg['rollout_dict']).eval()
    return (savi.to(dev), roll.to(dev)) if dev is not None else (savi, roll)


def synthetic_img(B, T, res, seed=1234):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.rand(B, T, 3, res, res) * 2 - 1).astype(np.float32))


def encode_flops_per_frame(savi, folded=False):
    """Algorithmic FLOPs of the reference's encode per frame (savi.py:220-250, 66-102, 367-416): CNN, per-pixel MLP, k / v
    projections, Slot-Attention iterations, slot update.  folded=True leaves out the k / v projection this library folds into
    project_q / the GRU input matrix (exact algebra; the work is gone, not moved)."""
    ch, ks = list(savi.enc_channels), savi.enc_ks
    HW = 64 * 64
    f = 0.0
    for i in range(len(ch) - 1):
        f += 2.0 * HW * ch[i + 1] * ch[i] * ks * ks
    Ce, D, N, Hm = savi.enc_out_channels, savi.slot_size, savi.num_slots, savi.slot_mlp_size
    f += 2.0 * HW * (ch[-1] * Ce + Ce * Ce)                      # encoder_out_layer
    if not folded:
        f += 2.0 * HW * Ce * D * 2                               # project_k, project_v
    it = savi.num_iterations
    f += it * (4.0 * HW * N * D)                                 # logits + weighted sum
    f += it * N * (2.0 * D * D + 2.0 * 3 * D * D * 2 + 4.0 * D * Hm)   # project_q, GRUCell, residual MLP
    return f


def rollout_flops(roll, T, H):
    """Algorithmic FLOPs of one video's H-step rollout as the reference computes it (slotformer.py:85-126,
    single_step_slotformer.py:49-90): in_proj of the whole window, every layer on every row, out_proj of the newest frame."""
    N, C_, d = roll.num_slots, roll.in_proj.in_features, roll.in_proj.out_features
    nl, ffn = len(roll.transformer_encoder.layers), roll.transformer_encoder.layers[0].linear1.out_features
    single = hasattr(roll, 'cond_len')
    W = roll.cond_len if single else roll.history_len
    tot = 0.0
    for s in range(H):
        nf = min(s + 1, W) if single else W
        L = nf * N
        tot += 2.0 * L * C_ * d + nl * (2.0 * L * d * 3 * d + 4.0 * L * L * d + 2.0 * L * d * d + 4.0 * L * d * ffn) + 2.0 * N * d * C_
    return tot


ALLOW_STALE_TRACE = '--allow-stale-trace' in sys.argv


def committed_profile(key):
    """Entry `key` of the newest committed PMC summary (profiles/r*_pmc_traffic.json); {} when absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return {}
    try:
        full = json.load(open(files[-1]))
        d = dict(full.get(key, {}))
        d['source'] = os.path.relpath(files[-1], ROOT)
        # the trace describes the kernels of ONE source tree: a summary taken from other sources must not price this run
        now = source_tree_hash()
        if full.get('source_tree') != now and not ALLOW_STALE_TRACE:
            # no number of a trace taken from OTHER kernel sources enters the line: the objects fall back to this run's HIP events
            return {'stale_trace': os.path.relpath(files[-1], ROOT), 'trace_source_tree': full.get('source_tree'), 'source_tree': now}
        d['source_tree'] = now
        return d
    except ValueError:
        return {}


KERNEL_CLASS = {'layer_tok_kernel': 'layer_tok', 'conv5x5_halo': 'conv', 'ffn_qkv_tile_kernel': 'ffn_fused', 'ffn_tile_kernel': 'ffn_fused', 'ffn_partial_kernel': 'ffn_fused', 'ffn64_parts_kernel': 'ffn_fused', 'ffn_wide_parts_kernel': 'ffn_fused',
                'conv5x5_rows4_kernel': 'conv', 'conv5x5_ws_kernel': 'conv', 'qkv_rows_kernel': 'attention', 'attn_core_kernel': 'attention', 'attn_oproj_kernel': 'attention', 'attn_all_kernel': 'attention', 'seam_kernel': 'seam', 'sa_attn_mfma_kernel': 'slot_attn', 'sa_attn_fold_kernel': 'slot_attn', 'sa_attn_tile_kernel': 'slot_attn', 'sa_attn_planes_kernel': 'slot_attn'}


def dominant_kernel():
    """The kernel class with the largest total device time in the newest committed rocprof summary of this command
    (profiles/r*_kernel_stats.csv) -> key of the roofline object that becomes `roofline`."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_stats.csv')))
    files = [f for f in files if 'training' not in f]
    if not files:
        return 'ffn_fused', None
    tot = {}
    try:
        for r in csv.DictReader(open(files[-1])):
            if 'deconv5x5s2' in r['Name'] or 'rows4_kernel<false, true>' in r['Name'] or 'decode_' in r['Name']:
                continue   # (the decode leg of the traced command: a secondary line, never the headline's dominant kernel)
            for pat, key in KERNEL_CLASS.items():
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


def cpu_baseline(cfg, B_full, sample_B):
    """The oracle (CPU port of the reference path, torch fp32) timed on the host cores on a bounded sample: sample_B videos of the
    same workload at the fastest thread count, plus one pass each on ONE thread and on the full batch."""
    import oracle
    scfg, rcfg, single, _, T, H, res, _ = cfg
    savi, roll = build_models(None, cfg)
    ssd = {k: v.detach() for k, v in savi.state_dict().items()}
    rsd = {'rollouter.' + k: v.detach() for k, v in roll.state_dict().items()}
    N, D = roll.num_slots, roll.in_proj.in_features
    steve = scfg['model'] == 'STEVE'

    def make(nv):
        img = synthetic_img(nv, T, res)
        noise = torch.randn(nv, T, N, D)

        def run():
            with torch.no_grad():
                if steve:
                    post = oracle.steve_encode(img, ssd, scfg)['slots']
                else:
                    post = oracle.savi_encode(img, ssd, scfg, noise=noise)['post_slots']
                fwd = oracle.single_step_rollouter_forward if single else oracle.rollouter_forward
                return fwd(post, H, rsd, rcfg['rollout_dict'])
        return run

    run = make(sample_B)
    # pick the thread count that is fastest on this host (more threads than ~32 only add synchronisation cost at these
    # sizes; 256 hardware threads ran >100x slower)
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
    while reps < 3 or (time.perf_counter() - t0 < 8.0 and reps < 40):
        run()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    fr = T + H
    # the full batch, one pass at the same thread count
    runB = make(B_full)
    t0 = time.perf_counter()
    runB()
    dtB = time.perf_counter() - t0
    # ONE thread: one pass over a quarter of the sample (bounded: a single core needs seconds per video)
    n1 = max(1, sample_B // 4)
    run1 = make(n1)
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    run1()
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(best_t)
    return dict(value=sample_B * fr / dt, unit='frames/s', cores=best_t, kind='port', cpu=cpu_model_string(),
                sample=f'oracle (torch-CPU fp32 restatement of the reference path), {sample_B} of {B_full} videos, '
                f'{reps} passes of encode {T} frames + {H}-step rollout, {dt:.2f} s per pass, '
                f'{best_t} threads (fastest of 8/16/32/64) on a {ncpu}-hardware-thread host ({cpu_model_string()})',
                full_batch={'value': B_full * fr / dtB, 'videos': B_full, 'cores': best_t, 'seconds': dtB, 'passes': 1},
                one_thread={'value': n1 * fr / dt1, 'videos': n1, 'cores': 1, 'seconds': dt1, 'passes': 1})


def fail(msg, args, code=0):
    """A run that cannot start: say why on stderr and as a JSON line without a value (never an assert)."""
    print(f'[bench] {msg}', file=sys.stderr, flush=True)
    if int(os.environ.get('RANK', 0)) == 0:
        print(json.dumps({'metric': 'rollout frames/sec', 'value': None, 'unit': 'frames/s', 'n_gpus': args.gpus, 'steps': args.steps,
                          'warmup': args.warmup, 'error': msg}), flush=True)
    sys.exit(code)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run (one rank per GPU over RCCL, rendezvous
    on 127.0.0.1).  Needs N visible devices; says so cleanly otherwise."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        print(f'[bench] --gpus {n} needs {n} devices, {have} visible on this node: nothing run', file=sys.stderr, flush=True)
        print(json.dumps({'metric': 'rollout frames/sec', 'value': None, 'unit': 'frames/s', 'n_gpus': n, 'devices_visible': have,
                          'error': f'needs {n} devices, {have} visible'}), flush=True)
        return 0
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != '--self-launch'] + ['--force-dist']   # (a launched job always forms its RCCL group, also with one rank)
    print(f'[bench] launching {n} ranks: {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def source_tree_hash():
    """sha256 over the kernel sources (slotformer_amd/csrc + the C-ABI header): ties a committed rocprof summary to the code it traced."""
    from slotformer_amd.build import source_tree_hash as h
    return h()


def decode_flops_per_slot_image(m):
    """Algorithmic FLOPs of the spatial-broadcast decoder per slot image as the reference computes it (savi.py:252-293,504-525): every
    transposed convolution 2 * out_pixels * Cout * Cin * k^2 / stride^2, + the 1x1 head (1.135 GFLOP at 128 x 128: SURVEY.md 8f N2)."""
    ch, ks, size = list(m.dec_channels), m.dec_ks, m.dec_resolution[0]
    f = 0.0
    for i in range(len(ch) - 1):
        st = m.decoder[i][0].stride[0]
        size *= st
        f += 2.0 * size * size * ch[i + 1] * ch[i] * ks * ks / (st * st)
    return f + 2.0 * size * size * ch[-1] * 4


def decode_leg(args, lib, savi, roll, ring, B, T, H, N, D, RES, dev, pipe, peak_chip, peak_note, world, elapsed_main):
    """Row N2 on the clock: the pipeline with its decode stage -- encode + rollout + decode of ALL predicted frames of every batch
    (reconstruction [B, H, 3, R, R] + uint8 postproc_mask segmentation [B, H, R, R]; per-slot tensors never materialised)."""
    from slotformer_amd import engine
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    pipe.close()
    with torch.no_grad():
        pd = EncodeRolloutPipeline(savi, roll, B, T, H, partition=pipe.partition if pipe.cu_split else 'none', group=pipe.G, decoder=savi)
        K = max(2, args.decode_steps)
        dec = {}
        batches = [ring[j % 3] for j in range(K)]
        out_d = torch.empty(K, B, T + H, N, D, device=dev)
        pd.run(batches[:max(2, min(K, 4))], None, out=out_d[:max(2, min(K, 4))], decoded=None)   # warm the two front stages
        pd.run(batches, None, out=out_d, decoded=dec)                                                # and the decode stage (workspaces)
        torch.cuda.synchronize()
        wins = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pd.run(batches, None, out=out_d, decoded=dec)
            torch.cuda.synchronize()
            wins.append(time.perf_counter() - t0)
        assert torch.isfinite(dec['recon']).all() and int(dec['seg'].max()) < N
        el = sorted(wins)[1]
        # the decode alone: 32 frames (one frame-chunk of the library), and one whole batch's predicted frames
        sl32 = out_d[0][:, T:].reshape(B * H, N, D)[:32].contiguous()
        slB = out_d[0][:, T:].reshape(B * H, N, D).contiguous()

        def timed(fn, n=5):
            fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]

        t32 = timed(lambda: engine.savi_decode(savi, sl32, want=('seg', ), seg_dtype=torch.uint8))
        t32_full = timed(lambda: engine.savi_decode(savi, sl32))           # with recons + masks (the module API's outputs)
        tB = timed(lambda: engine.savi_decode(savi, slB, want=('seg', ), seg_dtype=torch.uint8, out_recon=dec['recon'][0].view(B * H, 3, RES, RES),
                                              out_seg=dec['seg'][0].view(B * H, RES, RES)), n=3)
        # the head-fused last layer, event-timed (library brackets around eager launches; the decode stage is never graph-captured)
        lib.sf_profile_enable((1 << 8) | (1 << 0))
        read_profile(lib)
        engine.savi_decode(savi, slB, want=('seg', ), seg_dtype=torch.uint8)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        prof = read_profile(lib)
        pd.close()
    fl_img = decode_flops_per_slot_image(savi)
    frames = world * B * (T + H) * K
    obj = {
        'note': 'SECONDARY line, never `value`: encode + rollout + decode of every predicted frame (test_vp.py-shaped use); the decode is '
                f'{fl_img * N * H / 1e9:.0f} GFLOP per video against {(T * encode_flops_per_frame(savi) + rollout_flops(roll, T, H)) / 1e9:.0f} for encode + rollout',
        'frames_per_s': frames / el, 'ms_per_step': 1e3 * el / K, 'steps': K, 'windows_ms_per_step': [1e3 * x / K for x in wins],
        'decoded_frames_per_s': world * B * H * K / el, 'decoded_frames_per_step': B * H, 'resolution': RES,
        'outputs': f'recon [B,{H},3,{RES},{RES}] f32 + postproc_mask segmentation [B,{H},{RES},{RES}] uint8 per batch; recons / masks not materialised',
        'decode_alone_ms_32_frames': 1e3 * t32, 'decode_alone_ms_32_frames_with_recons_and_masks': 1e3 * t32_full,
        'decode_alone_ms_per_batch': 1e3 * tB, 'decode_alone_frames_per_s': B * H / tB,
        'decode_tflops': fl_img * N * B * H / tB / 1e12, 'decode_frac_of_roof': fl_img * N * B * H / tB / 1e12 / peak_chip,
        'vs_without_decode': {'frames_per_s': world * B * (T + H) * args.steps / elapsed_main},
        'schedule': 'decode of a unit on an unmasked stream of its own behind the unit\'s rollout (slotformer_amd/pipeline.py, decoder=...); '
                    'bit-identical to the serial module calls (tests/test_pipeline_gpu.py::test_pipeline_with_the_decode_stage)',
    }
    if args.config == 'C2' and RES == 128:
        # the same leg at the reference's own CLEVRER resolution (stosavi_clevrer_params.py:33: 64 x 64; the first convolution has stride 1 there and
        # the decoder ends in a stride-1 layer): a second StoSAVi of that resolution, the same rollouter
        from slotformer_amd import configs
        from slotformer_amd.base_slots import build_model
        c64 = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench_configs()['C2'][0].items()}
        c64['resolution'] = (64, 64)
        torch.manual_seed(0)
        savi64 = build_model(configs.ParamsView(c64)).eval().to(dev)
        savi64.testing = True
        ring64 = [synthetic_img(B, T, 64, seed=4321 + k).to(dev) for k in range(3)]
        with torch.no_grad():
            p64 = EncodeRolloutPipeline(savi64, roll, B, T, H, decoder=savi64)
            d64, o64 = {}, torch.empty(K, B, T + H, N, D, device=dev)
            b64 = [ring64[j % 3] for j in range(K)]
            p64.run(b64, None, out=o64, decoded=d64)
            torch.cuda.synchronize()
            w64, w64n = [], []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                p64.run(b64, None, out=o64, decoded=d64)
                torch.cuda.synchronize()
                w64.append(time.perf_counter() - t0)
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                p64.run(b64, None, out=o64)
                torch.cuda.synchronize()
                w64n.append(time.perf_counter() - t0)
            assert torch.isfinite(d64['recon']).all()
            p64.close()
        e64, e64n = sorted(w64)[1], sorted(w64n)[1]
        obj['at_64x64'] = {'frames_per_s': world * B * (T + H) * K / e64, 'ms_per_step': 1e3 * e64 / K, 'decoded_frames_per_s': world * B * H * K / e64,
                           'without_decode_frames_per_s': world * B * (T + H) * K / e64n, 'steps': K,
                           'note': 'the reference trains / evaluates CLEVRER at 64 x 64 (stosavi_clevrer_params.py:33): StoSAVi of that resolution (stride-1 first convolution, '
                                   'decoder 8 -> 16 -> 32 -> 64 + a stride-1 layer with the head in its epilogue), same rollouter, same schedule'}
    roof = None
    dh = prof.get('deconv_head')
    if dh:
        fl = dh['work'] / dh['launches']
        pm = committed_profile('deconv_head') if args.config == 'C2' else {}
        us_trace = pm.get('avg_launch_us_trace')
        us = us_trace or dh['avg_us']
        tf = fl / (us * 1e-6) / 1e12
        # slot images per launch: the layer reads a [64, 64, 64] map per image and writes 16 bytes per output pixel
        R_img = fl / (2.0 * 4096 * 64 * 64 * 25 + 2.0 * RES * RES * 64 * 4)
        roof = {'kernel': ('deconv5x5s2_kernel<64, true>' if RES == 128 else 'conv5x5_rows4_kernel<false, true>') +
                          ' (the last decoder layer: 5x5 transposed convolution 64 -> 64 to the output resolution + ReLU + the 1x1 head in the epilogue; ' + peak_note + ')',
                'bound': 'mfma', 'achieved': tf, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': tf / peak_chip,
                'flops_per_launch': fl, 'avg_launch_us': us, 'avg_launch_us_events': dh['avg_us'], 'avg_launch_us_rocprof': us_trace,
                'frac_events': fl / (dh['avg_us'] * 1e-6) / 1e12 / peak_chip, 'launches': dh['launches'],
                'frac_source': ('committed rocprof trace of this source tree (' + str(pm.get('source')) + ')') if us_trace else 'HIP events of this run (eager launches of one decode call)',
                'traffic': pm.get('traffic_bytes_per_launch'), 'mfma_busy_frac': pm.get('mfma_busy_frac'),
                'slot_images_per_launch': R_img, 'algorithmic_bytes_per_launch': R_img * (4096 * 64 * 4 + RES * RES * 16) + 409600,
                'share_of_decode_flops': 2.0 * RES * RES * 64 * 64 * 25 / (4 if RES == 128 else 1) / fl_img}
    return obj, roof


def log(msg):
    if int(os.environ.get('RANK', 0)) == 0:
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--config', choices=['C2', 'C4', 'C5'], default='C2', help='BASELINE.json configuration (default C2: the one the metric is quoted on)')
    ap.add_argument('--batch', type=int, default=None, help='videos per GPU (default: the configuration\'s batch)')
    ap.add_argument('--no-graph', action='store_true', help='launch the rollout eagerly instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample', type=int, default=8)
    ap.add_argument('--no-overlap', action='store_true', help='run encode and rollout of each batch back-to-back on one stream '
                    '(default: the batch pipeline of slotformer_amd/pipeline.py)')
    ap.add_argument('--precision', choices=['bf16x3', 'f32'], default=None, help='matrix arithmetic mode (default: library default = bf16x3)')
    ap.add_argument('--pcie', action='store_true', help='also time a host-to-host (PCIe-inclusive) variant; reported separately')
    ap.add_argument('--breakdown', action='store_true', help='extra untimed pass with every kernel class timed')
    ap.add_argument('--decode', action='store_true', help='also time encode + rollout + DECODE of every predicted frame (reconstruction + postproc_mask '
                    'segmentation, what test_vp.py scores): a secondary object `decode_pipeline` + `roofline_decode`, never `value`')
    ap.add_argument('--decode-steps', type=int, default=8, help='batches per timed window of the --decode leg')
    ap.add_argument('--group', type=int, default=None, help='batches per rollout unit of the pipeline (default: the pipeline\'s own choice)')
    ap.add_argument('--roll-streams', type=int, default=2, help="rollout streams of the 'pair' partition (units side by side on the rollout CUs; default 2)")
    ap.add_argument('--hybrid', type=int, default=None, help='every k-th batch behind the fill on an unmasked stream (default: the pipeline\'s own choice)')
    ap.add_argument('--enc-group', type=int, default=0, help='batches handed to the pipeline as one (default: pipeline.encode_group_for)')
    ap.add_argument('--cu-split', default='ff', help='encode CU mask of the pipeline: hex word, or rows<R> (pipeline.encode_mask_words)')
    ap.add_argument('--partition', choices=['pair', 'three', 'two', 'none'], default='pair')
    ap.add_argument('--steal', type=float, default=None, help='time steps of convolutions per batch computed on the rollout streams (default: the partition\'s)')
    ap.add_argument('--split', choices=['enc', 'roll'], default=None,
                    help='OPT-IN: the encode in two halves (csrc/slot_chain.hip): features on the encode lane, the slot branch of a rollout unit as one launch on the encode side (enc) or at the head of its rollout graph (roll)')
    ap.add_argument('--force-dist', action='store_true', help='form the RCCL process group with one rank too (the multi-GPU path on one GPU)')
    ap.add_argument('--self-launch', action='store_true', help='take the torch.distributed.run self-launch path with --gpus 1 too')
    ap.add_argument('--live-every', type=int, default=4, help='bracket every n-th conv / Slot-Attention launch with events in the untimed live pass')
    ap.add_argument('--live-mask', type=int, default=(1 << 0) | (1 << 3), help='kernel classes of that pass')
    ap.add_argument('--allow-stale-trace', action='store_true', help='use the committed rocprof summary although the kernel sources changed')
    ap.add_argument('--windows', type=int, default=5, help='timed windows (at least) of --steps steps each on the warm pipeline; value = the median window')
    ap.add_argument('--min-timed-s', type=float, default=1.0, help='keep adding windows until the timed region holds this many seconds (0: exactly --windows)')
    args = ap.parse_args()

    if (args.gpus > 1 or args.self_launch) and 'WORLD_SIZE' not in os.environ:
        # (--self-launch: take the launcher path with --gpus 1 too -- how tests/test_dist_gpu.py runs it on the one GPU of the box)
        # plain `python bench.py --gpus N`: this process becomes the launcher of its own N ranks (one process per GPU, the
        # reference's launch shape: scripts/sbatch_run.sh:36-42) and passes rank 0's JSON line through
        sys.exit(self_launch(args.gpus))
    # every SF_* variable of the environment must be one the library / pipeline / this script reads (slotformer_amd/switches.py); the ones that are set
    # go into the line (config.switches): a number measured with a probe switch on says so
    from slotformer_amd.switches import check_environment
    switches_set, switches_unknown = check_environment()
    if switches_unknown:
        fail(f'unknown SF_* environment variable(s) {switches_unknown}: not in slotformer_amd/switches.py (a typo would silently measure the defaults)', args, code=2)
    rank = int(os.environ.get('RANK', 0))
    pipe_closed = False
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        fail(f'--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree', args, code=2)
    if not torch.cuda.is_available():
        fail('bench.py needs a HIP device (none visible)', args, code=2)
    if torch.cuda.device_count() <= local:
        fail(f'rank {rank} needs device {local}, {torch.cuda.device_count()} visible', args, code=2)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or args.force_dist  # the latter: exercise the RCCL path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        # RCCL, its communicator created at the first collective (no device_id): the pipeline's streams get their hardware queues before RCCL's does
        # (worth 3 % on one rank: pipeline._pick_free_streams)
        dist.init_process_group('nccl', rank=rank, world_size=world)

    from slotformer_amd import engine, _lib
    from slotformer_amd.pipeline import EncodeRolloutPipeline
    lib = _lib.lib()
    if args.precision:
        lib.sf_set_precision(1 if args.precision == 'bf16x3' else 0)
    prec = 'bf16x3' if lib.sf_get_precision() == 1 else 'f32'
    cfg = bench_configs()[args.config]
    scfg, rcfg, single, B_def, T_BURN, T_ROLL, RES, workload = cfg
    B = args.batch or B_def
    savi, roll = build_models(dev, cfg)
    N_SLOTS, SLOT_D = roll.num_slots, roll.in_proj.in_features
    # a ring of three DIFFERENT resident inputs: consecutive batches never see the same frames
    ring = [synthetic_img(B, T_BURN, RES, seed=1234 + 1000 * k + rank).to(dev) for k in range(3)]
    # small batches are ENCODED several at a time (E consecutive batches concatenated into one encode launch set of ~32 videos): the slot branch of an
    # encode (predictor, Slot-Attention iterations, slot updates) is latency-bound and costs the same for 16 videos as for 32 (C4: 3.3 vs 2.5 ms per 16
    # videos on the encode lane).  Every batch still runs its full encode + rollout inside the timed region; the concatenation is part of it.
    from slotformer_amd.pipeline import encode_group_for
    E = int(args.enc_group) or encode_group_for(B, args.steps)
    if args.steps % E or args.no_overlap:
        E = 1
    Bp = B * E                                                   # videos per pipeline batch
    ringp = ring if E == 1 else [torch.cat([ring[(k + i) % 3] for i in range(E)], 0) for k in range(3)]   # (for the untimed extras)
    cu_word = args.cu_split          # hex mask word, or rows<R> (pipeline.encode_mask_words)
    cu_word = cu_word if cu_word.startswith('rows') else int(cu_word, 16)
    steal = args.steal                   # None: the partition's default
    steal = None if steal is None else float(steal)
    partition = args.partition   # 'pair' | 'three' | 'two' | 'none' (pipeline.EncodeRolloutPipeline)
    group = args.group
    group = None if group is None else int(group)
    if group is None:
        from slotformer_amd.pipeline import unit_batches_for
        group = unit_batches_for(roll, Bp, args.steps // E, T_BURN)   # (as harness.extract_and_rollout: larger units for long runs of small batches; None for C2 / C5)

    with torch.no_grad():
        log('building the pipeline (first eager rollouts + graph capture)')
        pipe = EncodeRolloutPipeline(savi, roll, Bp, T_BURN, T_ROLL, encode_cu_word=cu_word, steal_steps=steal,
                                     use_graph=not args.no_graph, partition=partition, group=group, split=bool(args.split), chain_on=args.split or 'enc',
                                     roll_streams=args.roll_streams, hybrid=args.hybrid)
        overlap = not args.no_overlap
        G = pipe.G                      # batches per rollout graph
        unit0 = pipe.units[0]
        graph = unit0.graph

        def run(n, out):   # n batches of B videos, E at a time
            if E == 1:
                return pipe.run([ring[j % 3] for j in range(n)], None, out=out, serial=not overlap)
            ns = max(1, n // E)
            return pipe.run([torch.cat([ring[(s * E + i) % 3] for i in range(E)], 0) for s in range(ns)], None,
                            out=out.view(-1, Bp, *out.shape[2:])[:ns], serial=not overlap)

        def barrier():
            if use_dist:
                dist.barrier(device_ids=[dev.index])
            torch.cuda.synchronize()

        shape = (B, T_BURN + T_ROLL, N_SLOTS, SLOT_D)
        out_w = torch.empty((max(args.warmup, 6) + E, ) + shape, device=dev)
        out_t = torch.empty((args.steps, ) + shape, device=dev)
        run(args.warmup, out_w)
        torch.cuda.synchronize()
        assert torch.isfinite(out_w[:args.warmup]).all(), 'non-finite slots in the warmup batches'
        unit_sizes = None
        if overlap:   # the graphs of the timed run's unit plan exist before the clock starts (a remainder's units are captured on first use otherwise)
            unit_sizes = pipe.prepare(max(1, args.steps // E) if E > 1 else args.steps)
            log(f'rollout units of the timed run: {unit_sizes}')
        log('warmup done')
        # the timed region carries no measurement probes: HIP-event brackets around the conv / Slot-Attention launches of the encode
        # stream cost that stream ~5 % even when only every 4th launch is bracketed (341 vs 359 k frames/s)
        # R timed windows on the same warm pipeline, each EXACTLY args.steps steps bracketed by barrier + synchronize on both
        # sides; `value` / `ms_per_step` come from the MEDIAN window, p10 / p90 over the windows beside it
        windows, unit_gaps = [], []
        # at least --windows windows, and (unless --windows was given) as many more as it takes for the timed region to reach --min-timed-s seconds:
        # the count is agreed between the ranks (rank 0's clock) so that every rank runs the same number of windows
        w = 0
        while True:
            if w >= max(1, args.windows):
                more = torch.tensor([1.0 if (sum(windows) < args.min_timed_s and w < 64) else 0.0], device=dev)
                if use_dist:
                    dist.broadcast(more, 0)
                if more.item() == 0.0:
                    break
            w += 1
            barrier()
            t0 = time.perf_counter()
            run(args.steps, out_t)
            barrier()
            windows.append(time.perf_counter() - t0)
            if overlap and len(pipe.completion_events) > 1:
                # device time between the completions of consecutive rollout units (units on the unmasked drain streams may
                # overtake their predecessors: the completion times are sorted first)
                ev0 = pipe.completion_events[0]
                done = sorted((ev0.elapsed_time(e), nb) for e, nb in zip(pipe.completion_events, pipe.completion_batches))
                unit_gaps += [(done[j][0] - done[j - 1][0]) / max(done[j][1] * E, 1) for j in range(1, len(done))]
        windows_all = list(windows)
        if use_dist:   # every window: the slowest rank's time
            tw = torch.tensor(windows, device=dev, dtype=torch.float64)
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            windows_all = tw.tolist()
        elapsed = sorted(windows_all)[len(windows_all) // 2]
        log(f'timed windows done: {[round(x, 4) for x in windows_all]} s')
        # conv + Slot-Attention launches LIVE: a second pass of the same schedule with library brackets (HIP events on the launch
        # stream) around every LIVE_EVERY-th launch of the two classes
        LIVE_EVERY = int(args.live_every)
        lib.sf_profile_sample(LIVE_EVERY)
        lib.sf_profile_enable(int(args.live_mask))
        read_profile(lib)
        run(max(args.warmup, 6), out_w)
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        lib.sf_profile_sample(1)
        prof = read_profile(lib)
        assert torch.isfinite(out_t).all(), 'non-finite slots in the timed batches'
        # the three ring inputs give three different results (a stale slot buffer would repeat one)
        if args.steps >= 3:
            assert not torch.equal(out_t[0, :, :T_BURN], out_t[1, :, :T_BURN])

        # ---- untimed extras: the two halves alone, per-kernel event timings ----
        noise = engine.kernel_noise(savi, None, B, T_BURN, dev)
        noise_p = engine.kernel_noise(savi, None, Bp, T_BURN, dev)

        def encode():
            post, _, _ = engine.savi_encode(savi, ring[0], noise=noise, side_stream=None)   # one stream: the kernels alone, for the rooflines
            unit0.buf[:B, :T_BURN].copy_(post)

        def encode_forked():   # the direct module path's default: two branches (engine.savi_encode side_stream='auto')
            engine.savi_encode(savi, ring[0], noise=noise)

        def rollout_eager():
            engine.rollout(roll, unit0.buf, T_BURN, T_ROLL, ws_slot=unit0.key, opts=pipe.rollout_opts)

        def rollout():          # one rollout UNIT: G batches
            graph.replay() if graph is not None else rollout_eager()

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

        # one batch at a time (no pipelining, whole chip): the latency of ONE batch of B videos through encode + rollout
        one_batch = []
        for k in range(6):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pipe.run([ringp[k % 3]], None, serial=True)
            torch.cuda.synchronize()
            one_batch.append(time.perf_counter() - t1)
        one_batch_ms = 1e3 * sorted(one_batch[1:])[len(one_batch[1:]) // 2]   # (the first call captures the one-batch unit's graph)
        torch.cuda.synchronize()
        # the same through the MODULE calls a reference script makes per batch (savi forward, then the rollouter: extract_slots.py:19-38 ->
        # rollout_clevrer_slots.py:20-65; engine.savi_encode in its default two-branch form + engine.rollout in the library's default forms), eager
        one_buf = torch.zeros(Bp, T_BURN + T_ROLL, N_SLOTS, SLOT_D, device=dev)
        direct = []
        for k in range(6):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            post_k, _, _ = engine.savi_encode(savi, ringp[k % 3], noise=noise_p)
            one_buf[:, :T_BURN].copy_(post_k)
            engine.rollout(roll, one_buf, T_BURN, T_ROLL, ws_slot=('bench', 'one'))
            torch.cuda.synchronize()
            direct.append(time.perf_counter() - t1)
        one_batch_direct_ms = 1e3 * sorted(direct[1:])[len(direct[1:]) // 2]
        lib.sf_profile_enable((1 << 0) | (1 << 3))
        t_enc = timed_on(torch.cuda.current_stream(), encode)
        lib.sf_profile_enable(0)
        prof_iso = read_profile(lib)  # same kernels with nothing else on the GPU
        t_enc_fork = timed_on(torch.cuda.current_stream(), encode_forked)
        t_roll = timed_on(torch.cuda.current_stream(), rollout)      # seconds per unit (G batches)
        part_ms = None
        if overlap and pipe.cu_split:
            def lane_encode(li):
                st, lo, hi = pipe.lanes[li]
                return lambda: pipe._encode(ringp[0], noise_p, unit0.buf[:Bp], None, lo, hi, li, unit=(unit0, 0))

            part_ms = {'encode_lane_ms_on_its_cus': [round(1e3 * timed_on(pipe.lanes[li][0], lane_encode(li)), 4) for li in range(len(pipe.lanes))],
                       'encode_lane_videos': [hi - lo for _, lo, hi in pipe.lanes],
                       'rollout_unit_ms_on_its_cus': 1e3 * timed_on(pipe.s_roll, rollout), 'batches_per_rollout_unit': G * E}
            part_ms['encode_ms_on_its_cus'] = max(part_ms['encode_lane_ms_on_its_cus'])
        # the rollout layer kernels, event-timed: (a) alone on the whole chip in one eager rollout unit, (b) LIVE in a pipelined
        # pass with the product schedule (encode stream busy on its CUs) but eager launches -- inside the timed region they
        # replay from a hipGraph, where HIP events cannot be inserted between the kernels
        lib.sf_profile_enable((1 << 5) | (1 << 6) | (1 << 7) | (1 << 9))
        read_profile(lib)
        rollout_eager()
        torch.cuda.synchronize()
        prof_roll = read_profile(lib)
        prof_roll_live = {}
        if overlap:
            saved = [(u, u.graph) for u in pipe.units]
            for u in pipe.units:
                u.graph = None
            pipe.run([ringp[j % 3] for j in range(4 * G)], None)
            torch.cuda.synchronize()
            for u, g in saved:
                u.graph = g
            prof_roll_live = read_profile(lib)
        lib.sf_profile_enable(0)
        breakdown = None
        if args.breakdown:
            lib.sf_profile_enable(0x3ff)
            encode()
            rollout_eager()
            torch.cuda.synchronize()
            lib.sf_profile_enable(0)
            breakdown = read_profile(lib)

    per_rank = None
    rccl_ranks, rccl_backend = None, None
    if use_dist:
        # per-rank median window (the line's value uses the max over ranks of every window) + what the process group really is
        mine = torch.tensor([sorted(windows)[len(windows) // 2]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [B * (T_BURN + T_ROLL) * args.steps / x.item() for x in allr]
        rccl_ranks, rccl_backend = dist.get_world_size(), str(dist.get_backend())

    if rank == 0:
        frames = world * B * (T_BURN + T_ROLL) * args.steps
        step_dist = None
        pick_of = lambda v, q: sorted(v)[min(len(v) - 1, int(q * len(v)))]  # noqa: E731
        if len(unit_gaps) > 2:
            step_dist = {'p10': pick_of(unit_gaps, 0.10), 'median': pick_of(unit_gaps, 0.50), 'p90': pick_of(unit_gaps, 0.90), 'n': len(unit_gaps),
                         'batches_per_unit': G,
                         'note': 'device time between the SORTED completion times of consecutive rollout units / batches per unit, all windows (rank 0)'}
        peak_chip = PEAK_BF16_MFMA_TFLOPS / 3.0 if prec == 'bf16x3' else PEAK_F32_MFMA_TFLOPS
        peak_note = ('split-bf16 MFMA: 3 bf16 MFMA flops per algorithmic flop -> roof = 2500/3 TFLOP/s' if prec == 'bf16x3'
                     else 'exact f32 MFMA')
        nl = len(roll.transformer_encoder.layers)
        enc_f, enc_f_fold = encode_flops_per_frame(savi), encode_flops_per_frame(savi, folded=True)
        roll_f = rollout_flops(roll, T_BURN, T_ROLL)             # per video
        step_flops = B * (T_BURN * enc_f + roll_f)
        step_flops_fold = B * (T_BURN * enc_f_fold + roll_f)
        metric = ('rollout frames/sec at B=32, 128x128, 7 slots, 6+50 steps (SAVi-encode + SlotFormer rollout)' if args.config == 'C2' and B == 32 else
                  f'rollout frames/sec, config {args.config} at B={B}, {RES}x{RES}, {N_SLOTS} slots, {T_BURN}+{T_ROLL} steps (slot extraction + SlotFormer rollout)')
        res = {
            'metric': metric,
            'value': frames / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_step_windows': {'median': 1e3 * elapsed / args.steps, 'p10': 1e3 * pick_of(windows_all, 0.10) / args.steps,
                                    'p90': 1e3 * pick_of(windows_all, 0.90) / args.steps, 'n': len(windows_all),
                                    'all': [1e3 * x / args.steps for x in windows_all],
                                    'note': 'windows of exactly `steps` steps each on the same warm pipeline, each bracketed by barrier + synchronize; '
                                            'value / ms_per_step = the median window (max over ranks per window)'},
            'timed_region_s': sum(windows_all),
            'per_rank_frames_per_s': per_rank, 'rccl_ranks': rccl_ranks, 'process_group_backend': rccl_backend,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 storage; matmul/conv on split-bf16 MFMA (bf16x3: hi*hi+hi*lo+lo*hi, f32 accumulate)' if prec == 'bf16x3' else 'f32',
            'data': 'synthetic',
            'config': {
                'workload': workload + '; random-init weights',
                'name': args.config,
                'batch_per_gpu': B, 'batches_per_encode': E, 'global_batch': B * world, 'burn_in': T_BURN, 'rollout': T_ROLL,
                'parallelism': f'dp{world}: videos sharded on the batch axis, no collective on the timed path',
                'inputs': 'ring of 3 different resident batches; the slots of every batch are copied out of the slot buffers',
                'schedule': 'slotformer_amd.pipeline.EncodeRolloutPipeline (product code, tests/test_pipeline_gpu.py)',
                'stream_placement': getattr(pipe, 'stream_placement', None),
                'switches': switches_set,   # the SF_* environment variables set for this run ({}: every default)
                'rollout_form': ('token-stationary layer launches (csrc/layer_tok.hip) for the layers before the last in the full units; row-tile forms for '
                                 'the last layer, the last unit of a run and remainder units; latency forms below 96 videos') if getattr(pipe, 'tok', False) else
                                'row-tile / latency forms (csrc/attn_rows.hip, ffn_tile.hip, layer_fused.hip)',
                'rollout_launch': (f'hipGraph replay, one graph per rollout unit of {G * E} batch(es) = {G * Bp} videos') if graph is not None else 'eager',
                'rollout_units_of_the_timed_run': ([u * E for u in unit_sizes] if overlap else None),   # batches per unit (the last two units take the remainder: pipeline.unit_sizes_for)
                'rollout_opts': None if pipe.rollout_opts is None else {k: getattr(pipe.rollout_opts, k) for k, _ in pipe.rollout_opts._fields_},
                'pipelining': ('encode of later batches (stream A) overlaps the rollout graphs of earlier units (streams B, C); every batch still '
                               'runs its full encode + rollout inside the timed region') if overlap else 'none',
                'work_stealing': (f'the CNN features of the first {pipe.steal:g} time step(s) (average) of a batch are computed on a rollout stream, '
                                  f'behind the rollout of the unit {pipe.lead} units earlier') if (overlap and pipe.steal) else 'none',
                'cu_partition': ((f'two rollout streams (two units roll out side by side) on {pipe.rollout_cus} CUs (whole CU rows of all four shader '
                                  f'engines of every XCD), the encode stream on the other {pipe.encode_cus}') if pipe.partition == 'pair' else
                                 ('rollout stream: CU rows 0-6 of shader engines 1-3 of every XCD (168 CUs); encode lane 0: shader engine 0 '
                                  f'(64 CUs, {pipe.lanes[0][2] - pipe.lanes[0][1]} videos of every batch); encode lane 1: CU row 7 of shader engines 1-3 '
                                  f'(24 CUs, {pipe.lanes[-1][2] - pipe.lanes[-1][1]} videos)') if pipe.partition == 'three' else
                                 (f'encode stream on CU mask {cu_word if isinstance(cu_word, str) else hex(cu_word)} ({pipe.encode_cus} CUs, the same '
                                  'number in every XCD), rollout stream on the complement')) if (overlap and pipe.cu_split) else 'none',
            },
            'one_batch_latency_ms': one_batch_ms,   # pipeline object, one batch per run() call (graph replays, one stream)
            'one_batch_latency_module_calls_ms': one_batch_direct_ms,   # savi forward + rollouter forward per batch, as the reference's scripts call them
            'one_batch_frames_per_s': Bp * (T_BURN + T_ROLL) / (one_batch_ms * 1e-3),
            'encode_ms': 1e3 * t_enc,
            'encode_ms_two_branches': 1e3 * t_enc_fork,
            'rollout_ms': 1e3 * t_roll / (G * E),
            'rollout_unit_ms': 1e3 * t_roll,
            'partitioned_ms': part_ms,
            'ms_per_step_distribution': step_dist,
            'encoded_frames_per_s': B * T_BURN / t_enc,
            'predicted_frames_per_s': G * Bp * T_ROLL / t_roll,
            'whole_step_tflops': step_flops * args.steps * world / elapsed / 1e12,
            'whole_step_flops': step_flops,
            'whole_step_flops_without_folded_kv_projection': step_flops_fold,
            'whole_step_tflops_without_folded_kv_projection': step_flops_fold * args.steps * world / elapsed / 1e12,
            'flops_per_encoded_frame': enc_f, 'flops_per_video_rollout': roll_f,
        }
        # ---- roofline objects, one per hot kernel; `roofline` = the one that dominates the committed rocprof summary of this
        #      command (profiles/r*_kernel_stats.csv), the others stay as secondary keys ----
        objs = {}
        W_FR = W_FR_ = roll.cond_len if single else roll.history_len
        tile_forms = pipe.rollout_opts is not None and pipe.rollout_opts.ffn_tile >= 1
        rows_forms = pipe.rollout_opts is not None and pipe.rollout_opts.attn_qkv_rows == 128
        fused_next = pipe.rollout_opts is not None and pipe.rollout_opts.ffn_tile == 2 and rows_forms
        tok_forms = pipe.rollout_opts is not None and pipe.rollout_opts.layer_tok > 0
        for key, name in (('layer_tok', f'layer_tok_kernel: the {nl - 1} layers before the last of a rollout step in ONE token-stationary launch -- per layer LN1, q|k|v, scores, '
                                        'softmax, PV, out-projection, LN2, FFN1, ReLU, FFN2 as one chain of register-resident split-bf16 MFMA products per 32-token wave (the '
                                        'accumulator layout of a product is the B operand of the next), a 128-token workgroup = 128 // L whole videos on one CU, the '
                                        f'weight fragments ({nl - 1} x 3 MB) streamed global -> LDS once per workgroup; {-(-G * Bp // max(128 // (W_FR_ * N_SLOTS), 1))} workgroups per unit launch '
                                        '(two units side by side fill the rollout CUs)'),
                          ('ffn_fused', ('ffn_qkv_tile_kernel (LN2 + FFN1 + ReLU + FFN2 over all four hidden chunks on a 64-row tile of finished rows AND LN1 + q|k|v of the '
                                         'NEXT layer on the same rows, weights streamed as MFMA fragments; fragment planes + parked residual rows out: flops_per_launch '
                                         'counts both) on the layers before the last, ffn_partial_kernel<1> (32-row tile x 256-wide hidden chunk, last-arriver reduction + '
                                         'step boundary) on the last layer') if fused_next else
                           ('ffn_tile_kernel (LN2 + FFN1 + ReLU + FFN2 over all four hidden chunks on a 64-row tile of finished rows, weights streamed as '
                                         'MFMA fragments; finished rows out) on the layers before the last, ffn_partial_kernel<1> (32-row tile x 256-wide hidden chunk, '
                                         'last-arriver reduction + step boundary) on the last layer') if tile_forms else
                           ('ffn_wide_parts_kernel<2> / <1> / ffn_partial_kernel (sum of the 4 head-pair partials + LN2 + FFN1 + ReLU + FFN2 on a '
                            '128- / 64- / 32-row tile x 256-wide hidden chunk; chunk partials out, last-arriver reduction + step boundary on the last layer)')),
                          ('attention', ('attn_core_kernel (one workgroup per video, wave = head: softmax(qk^T)v in registers from the fragment planes, out-projection + '
                                         'residual: finished rows) -- alone on the layers whose q|k|v planes the previous FFN launch wrote, behind qkv_rows_kernel (LN1 + q|k|v '
                                         'on 64-row tiles of the unit) on layer 0; flops_per_launch = mean over the blocks of a step (one with its projection, three without)') if fused_next else
                           ('qkv_rows_kernel (LN1 + q|k|v of all heads on 64-row tiles of the whole unit, weights streamed as MFMA fragments) + '
                                         'attn_core_kernel (one workgroup per video, wave = head: softmax(qk^T)v in registers, out-projection + residual: finished '
                                         'rows) -- TWO launches per attention block, timed and counted as one') if rows_forms else
                           ('attn_all_kernel (LN1 + q|k|v + softmax(qk^T)v of all 8 heads + out-projection + residual: finished rows; one '
                            'workgroup per video) in the throughput units; attn_oproj_kernel (one workgroup per head pair and video, four partials) '
                            'in the latency-form drain unit')),
                          ('seam', 'seam_kernel (last-layer FFN + step boundary of step s and the layer-0 attention of step s+1 in one grid, '
                           'tile-local write-through hand-off)')):
            iso, live = prof_roll.get(key), prof_roll_live.get(key)
            if not iso:
                continue
            fl = iso['work'] / iso['launches']
            pm = committed_profile(key) if args.config == 'C2' else {}
            us_trace = pm.get('avg_launch_us_trace')
            us_ev = (live or iso)['avg_us']          # measured in THIS run: HIP events around the launches, live in the pipeline where it has them
            us = us_ev
            tf = fl / (us * 1e-6) / 1e12
            tf_trace = fl / (us_trace * 1e-6) / 1e12 if us_trace else None
            objs[key] = {
                'kernel': name, 'bound': 'mfma', 'achieved': tf, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': tf / peak_chip,
                'peak_note': peak_note + '; whole-chip roof',
                # `frac` / `achieved` / `avg_launch_us`: this run's HIP events (eager launches beside the running pipeline); the committed rocprof trace of
                # graph replays beside them (`*_rocprof`)
                'frac_rocprof': tf_trace / peak_chip if tf_trace else None, 'achieved_rocprof': tf_trace,
                'frac_dominant_kernel': tf / peak_chip if key == 'layer_tok' else None,
                'flops_per_launch': fl, 'avg_launch_us': us,
                'measured': ('flops_per_launch (mean over the launches of one rollout unit, library accounting) / avg_launch_us_rocprof = the mean graph-replay '
                             f'duration of this kernel class in the committed rocprofv3 --kernel-trace of this command ({pm.get("source")}); '
                             'avg_launch_us_events = HIP events around eager launches in this run') if us_trace else
                            'HIP events around eager launches in this run (no committed rocprof trace for this configuration)',
                'avg_launch_us_rocprof': us_trace,
                'avg_launch_us_events': (live or iso)['avg_us'], 'avg_launch_us_events_isolated': iso['avg_us'],
                'frac_events': fl / ((live or iso)['avg_us'] * 1e-6) / 1e12 / peak_chip,
                'frac_events_isolated': fl / (iso['avg_us'] * 1e-6) / 1e12 / peak_chip,
                'frac_source': ('committed rocprof trace of this source tree (' + str(pm.get('source')) + ')') if us_trace else
                               ('HIP events of this run' + (f" (the committed trace {pm['stale_trace']} is of source tree {pm['trace_source_tree']}, "
                                                            f"this run's is {pm['source_tree']}: not used)" if pm.get('stale_trace') else '')),
                'source_tree': pm.get('source_tree') or source_tree_hash(),
                'launches_per_unit': iso['launches'], 'rows_per_launch_full_window': G * Bp * W_FR * N_SLOTS,
                'cus_available': pipe.rollout_cus if pipe.cu_split else 256,
                **({'workgroups_per_launch': -(-G * Bp // max(128 // (W_FR * N_SLOTS), 1)),
                    'frac_of_occupied_cus': tf / (peak_chip * min(256, -(-G * Bp // max(128 // (W_FR * N_SLOTS), 1))) / 256.0),
                    'frac_of_occupied_cus_rocprof': (tf_trace / (peak_chip * min(256, -(-G * Bp // max(128 // (W_FR * N_SLOTS), 1))) / 256.0)) if tf_trace else None}
                   if key == 'layer_tok' else {}),
                'traffic': pm.get('traffic_bytes_per_launch'),
                'mfma_busy_frac': pm.get('mfma_busy_frac'), 'pmc_source': pm.get('source'),
                'limiter': ('one wave per SIMD issuing ~4900 MFMAs per layer at ~45 cycles each against the pipe\'s 32: per 32 KB weight stage (48 MFMAs) one workgroup barrier, '
                            'the LDS latency of its first fragments, eight LDS-DMA issues (~40 cycles each) and the softmax / conversion VALU work of the neighbouring head '
                            'between the MFMAs; a launch occupies ceil(videos / 3) CUs -- 64 of 256 for a unit of 192 videos -- so its whole-chip fraction is bounded by 0.25 '
                            '(frac_of_occupied_cus prices the CUs it holds) (DESIGN.md 4, 5)') if key == 'layer_tok' else
                           ('row-tile forms: the matrix pipe inside the streamed products (q|k|v: 5.2 us per 128 rows x 256 columns = the MFMA bound; FFN: 7.1 us '
                            'per 64 rows x hidden chunk against 5.1), the un-overlapped ingest + LayerNorm in front of them (8 / 3.5 us) and the plane / row '
                            'stores behind them; attention core: 9.5 us per video around 2 us of MFMAs (fragment + weight ingest) (DESIGN.md 4, 5)') if tile_forms else
                           ('FFN: per-CU ingest of weight fragments + rows (a CU takes in 70-100 GB/s) around ~10 us of MFMAs per 128-row workgroup; '
                            'attention: a serial eight-wave workgroup per video -- per head pair 1.9 us of projection MFMAs inside 8.8 us of LDS '
                            'round trips, softmax and barriers (DESIGN.md 4, 5)'),
            }
        roll_flops = roll_f * G * Bp
        res['roofline_rollout_graph'] = {
            'kernel': f'hipGraph of one rollout unit ({G * E} batches, {G * Bp} videos): ring init + per step {nl} x (attention + out-proj partials, fused FFN)',
            'bound': f'latency ({sum(v["launches"] for v in prof_roll.values())} dependent launches, {G * Bp} videos per launch); MFMA roof shown for scale',
            'achieved': roll_flops / t_roll / 1e12, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': roll_flops / t_roll / 1e12 / peak_chip,
            'ms': 1e3 * t_roll, 'us_per_step': 1e6 * t_roll / T_ROLL, 'seam_timeouts': int(lib.sf_seam_timeouts()),
        }
        # (the encode of the timed schedule replays from a hipGraph: the library's event brackets see nothing there -- `live` then comes from
        #  the committed trace: the mean duration on the HIP queue with the MOST launches of the kernel = the CU-masked encode stream)
        conv = prof.get('conv_nhwc_implicit_gemm')
        iso = prof_iso.get('conv_nhwc_implicit_gemm')
        if conv or iso:
            pm = committed_profile('conv_nhwc_implicit_gemm') if args.config == 'C2' else {}
            live_src = 'HIP events (library brackets) in an eager pass of the timed schedule'
            if pm.get('avg_launch_us_trace_encode_queue') and iso:
                # (the committed trace of THIS source tree: the graph-replayed launches of the timed schedule on the slowest encode queue -- the eager event
                #  pass runs without the rollout graphs beside it and at a higher clock: 404 us against 520-610 in the trace)
                conv = {'work': iso['work'], 'launches': iso['launches'], 'avg_us': pm['avg_launch_us_trace_encode_queue']}
                live_src = 'mean duration on the slowest encode queue of the committed rocprof trace (' + str(pm.get('source')) + ')'
            elif not conv:
                conv = {'work': iso['work'], 'launches': iso['launches'], 'avg_us': iso['avg_us']}
            flops_live = conv['work'] / conv['launches']        # mean over the lanes' launches (each a share of the batch)
            ach = flops_live / (conv['avg_us'] * 1e-6) / 1e12
            # CUs one live launch runs on: the lanes work side by side, each on its own CUs
            enc_cus = pipe.encode_cus / len(pipe.lanes) if (overlap and pipe.cu_split) else 256
            flops_per_launch = iso['work'] / iso['launches'] if iso else flops_live
            us_trace = pm.get('avg_launch_us_trace')
            us = us_trace or (iso['avg_us'] if iso else None)
            ach_iso = flops_per_launch / (us * 1e-6) / 1e12 if us else None
            objs['conv'] = {
                'kernel': ('conv5x5_ws_kernel (weights stationary in registers, one four-wave workgroup per CU walking its rows through a ring of halo rows: the CU-masked '
                           'encode lane and the fill / hybrid graphs) and conv5x5_rows4_kernel (4-row tiles, streamed weight fragments: plain streams) -- all '
                           'time steps of a batch per launch' if prec == 'bf16x3' else 'sf_gemm_kernel<128,64,...,conv_nhwc>') + ' (5x5 conv 64->64 @64x64; ' + peak_note + ')',
                'bound': 'mfma (weights-stationary kernel: per output row and wave 300 MFMAs in ~11.2 k cycles against 9.6 k of pipe time -- two workgroup barriers, '
                         'the exchange of the cin halves through LDS and the fill / epilogue instructions between the MFMAs; the chip clocks ~2.1-2.2 GHz under this '
                         'load, the roof is priced at 2.4; DESIGN.md round 5)',
                'achieved': ach_iso, 'peak': peak_chip, 'unit': 'TFLOP/s', 'frac': (ach_iso / peak_chip) if ach_iso else None,
                'avg_launch_us': us, 'avg_launch_us_rocprof': us_trace, 'avg_launch_us_events_isolated': iso['avg_us'] if iso else None,
                'measured': 'flops_per_launch / avg_launch_us_rocprof (the whole-chip launches of the committed rocprof trace, default queue); '
                            '`live` = the same kernel on the encode partition beside the rollouts: HIP events (library brackets, every '
                            f'{LIVE_EVERY}th launch) in a second pass of the timed schedule when the encode launches eagerly, else the mean duration on the '
                            'CU-masked encode queue of the committed trace (avg_launch_us_trace_encode_queue)',
                'live': {'achieved': ach, 'cus': enc_cus, 'avg_launch_us': conv['avg_us'], 'launches': conv['launches'],
                         'flops_per_launch': flops_live, 'frac_of_partition_peak': ach / (peak_chip * enc_cus / 256.0), 'source': live_src,
                         'note': 'a pipelined pass of the timed schedule, beside the rollout graphs; stolen convolutions (rollout streams) are not part of this average'},
                'traffic': pm.get('traffic_bytes_per_launch'), 'mfma_busy_frac': pm.get('mfma_busy_frac'), 'pmc_source': pm.get('source'),
                'traffic_unit': 'bytes/launch (rocprofv3 PMC 2*FETCH_SIZE+WRITE_SIZE)',
                'algorithmic_bytes_per_launch': 2 * (flops_per_launch / (2.0 * 4096 * 64 * 1600)) * 4096 * 64 * 4 + 64 * 1600 * 4, 'flops_per_launch': flops_per_launch,
                'frames_per_launch': flops_per_launch / (2.0 * 4096 * 64 * 1600),
            }
        sa = prof.get('slot_attn_iter')
        iso = prof_iso.get('slot_attn_iter')
        if not sa and iso:
            pm = committed_profile('slot_attn_iter') if args.config == 'C2' else {}
            sa = {'work': iso['work'], 'launches': iso['launches'], 'avg_us': pm.get('avg_launch_us_trace_encode_queue') or iso['avg_us']}
        if sa:
            ub = iso['work'] / iso['launches'] if iso else sa['work'] / sa['launches']   # unique bytes (library accounting: k == v counted once)
            pm = committed_profile('slot_attn_iter') if args.config == 'C2' else {}
            us_trace = pm.get('avg_launch_us_trace')
            us = us_trace or (iso['avg_us'] if iso else None)
            kv_bytes = 2.0 * B * 4096 * SLOT_D * 4
            objs['slot_attn'] = {
                'kernel': 'Slot-Attention iteration (logits, softmax over slots, per-slot weighted sums) over the normalised pixel features -- keys and '
                          'values are the SAME array after the k / v fold (savi.py:66-89)', 'bound': 'hbm',
                'unique_bytes_per_launch': ub, 'bytes_per_launch': ub,
                'achieved': ub / (us * 1e-6) / 1e9 if us else None, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s',
                'frac': (ub / (us * 1e-6) / 1e9 / PEAK_HBM_GBPS) if us else None,
                'frac_unique': (ub / (us * 1e-6) / 1e9 / PEAK_HBM_GBPS) if us else None,
                'frac_vs_reference_kv_bytes': (kv_bytes / (us * 1e-6) / 1e9 / PEAK_HBM_GBPS) if us else None,
                'reference_kv_bytes_per_launch': kv_bytes,
                'avg_launch_us': us, 'avg_launch_us_rocprof': us_trace, 'avg_launch_us_events_isolated': iso['avg_us'] if iso else None,
                'traffic': pm.get('traffic_bytes_per_launch'), 'pmc_source': pm.get('source'),
                'live': {'achieved': sa['work'] / sa['launches'] / (sa['avg_us'] * 1e-6) / 1e9, 'avg_launch_us': sa['avg_us'], 'launches': sa['launches'],
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
            res['kernel_breakdown_one_unit'] = breakdown
        if args.decode and (not hasattr(savi, 'decoder') or E != 1):
            res['decode_pipeline'] = {'note': 'not run: --decode times the SAVi spatial-broadcast decoder stage (StoSAVi configurations, one batch per encode); '
                                              'STEVE decodes through its dVAE + Transformer decoder (tools/bench_steve_decoder.py)'}
        elif args.decode:
            res['decode_pipeline'], res['roofline_decode'] = decode_leg(args, lib, savi, roll, ring, B, T_BURN, T_ROLL, N_SLOTS, SLOT_D, RES, dev, pipe,
                                                                        peak_chip, peak_note, world, elapsed)
            pipe_closed = True
        pcie = None
        if args.pcie:
            # (this process's own pipeline object is closed first: with two pipelines alive -- twice the CU-masked queues -- the
            #  same call takes 120 instead of 94 ms)
            if not pipe_closed:
                pipe.close()
            pipe_closed = True
            # PCIe-inclusive variant (never `value`): frames start in pinned host memory and the slots end there
            from slotformer_amd import harness
            nv = args.steps * B
            vids_h = torch.cat([ring[j % 3].cpu() for j in range(args.steps)], 0).pin_memory()
            # warm-up with the same shapes: the pipeline is captured and kept, and the pinned output block of this size sits in
            # PyTorch's host allocator cache afterwards (a fresh 128 MB hipHostMalloc costs tens of ms)
            for _ in range(2):   # (the second call still allocates a second pinned block while the first result is alive)
                warm = harness.extract_and_rollout(savi, roll, vids_h, T_ROLL, batch_size=B, to_host=True)
            del warm
            times = []
            for _ in range(4):   # calls 3..6 with these shapes: the median (the third is still ~5 % off the steady state, tools/pcie_probe.py)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                out_h = harness.extract_and_rollout(savi, roll, vids_h, T_ROLL, batch_size=B, to_host=True)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t1)
            t_p = sorted(times)[len(times) // 2]
            assert out_h.shape[0] == nv and not out_h.is_cuda
            pcie = {'frames_per_s_host_to_host': nv * (T_BURN + T_ROLL) / t_p, 'frames_per_s_host_to_host_calls_3_to_6': [nv * (T_BURN + T_ROLL) / x for x in times], 'frames_per_s_device_resident': world * B * (T_BURN + T_ROLL) * args.steps / elapsed,
                    'h2d_bytes_per_batch': ring[0].numel() * 4, 'd2h_bytes_per_batch': int(np.prod(shape)) * 4,
                    'note': 'harness.extract_and_rollout(to_host=True): frames in pinned host memory, uploaded batch by batch on a copy stream ahead of '
                            'the encode, slots downloaded behind the rollout; median of calls 3..6 with these shapes (pipeline graphs and the pinned output blocks cached; tools/pcie_probe.py prints every call)'}
        if pcie:
            res['pcie_inclusive'] = pcie
        if world == 1 and not args.no_cpu_baseline:
            log('cpu baseline ...')
            res['cpu_baseline'] = cpu_baseline(cfg, B, min(args.cpu_sample, B))
        # RCCL prints its version banner through C stdio (flushed at exit when stdout is a pipe): push it out first so
        # that the JSON line is the last thing on stdout
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)
    if not pipe_closed:
        pipe.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
