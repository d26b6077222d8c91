"""RCCL on the GPU box, every round, although only one GPU is available: the process-group path of bench.py
(`bench.py --force-dist`: init + barrier + max-reduce of the elapsed time) and the flat-bucket gradient all-reduce of the
training path under a 1-rank `nccl` group (scripts/sbatch_run.sh:36-42 launches the reference the same way, one process per
GPU).  A 1 -> 8 GPU curve cannot be measured here; see DESIGN.md 6."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_under_rccl_process_group(dev):
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--force-dist'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['value'] > 0 and d['scaling'] == 'weak'
    assert 'roofline' in d and d['roofline']['frac'] <= 1.0


def test_bench_launches_its_own_ranks(dev):
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run (one rank per GPU, RCCL): the same code path
    with N = 1 (--self-launch) on the one GPU of the box -- the JSON line comes from rank 0 of the launched job with the process-group keys;
    and `--gpus 2` on this box fails cleanly with a JSON line that says what is missing (exit code 0, no value)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', '--windows', '2', '--min-timed-s', '0',
                        '--no-cpu-baseline', '--self-launch'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['rccl_ranks'] == 1 and 'nccl' in d['process_group_backend']
    assert len(d['per_rank_frames_per_s']) == 1 and d['ms_per_step_windows']['n'] == 2 and d['config']['stream_placement']['rccl_initialised']
    if torch.cuda.device_count() < 2:
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
        assert r2.returncode == 0
        d2 = json.loads([ln for ln in r2.stdout.strip().splitlines() if ln.startswith('{')][-1])
        assert d2['value'] is None and 'needs 2 devices' in d2['error']


def test_ddp_flat_bucket_under_one_rank_nccl(dev):
    """SlotFormer training with ddp_flat_bucket=True inside a 1-rank RCCL group: the in-backward all-reduce runs on the real
    backend and leaves the gradients unchanged (mean over one rank)."""
    import torch.distributed as dist
    from test_engine_gpu import build
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        cfg = gu.TRAIN_ROLL
        m, _ = build(cfg, gu.load_golden('roll_train'), 801, dev, vp=True)
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.
        rd = cfg['rollout_dict']
        slots = gu.seeded_normal((2, rd['history_len'] + cfg['loss_dict']['rollout_len'], rd['num_slots'], rd['slot_size']), 5).to(dev)

        def grads(flag):
            m.rollouter.ddp_flat_bucket = flag
            m.zero_grad(set_to_none=True)
            out = m({'slots': slots})
            m.calc_train_loss({'slots': slots}, out)['slot_recon_loss'].backward()
            return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}

        g0, g1 = grads(False), grads(True)
        assert g0.keys() == g1.keys() and len(g0) > 10
        for n in g0:
            assert torch.equal(g0[n], g1[n]), n
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)
        dist.barrier()
        assert torch.equal(t.cpu(), torch.ones(4))
    finally:
        dist.destroy_process_group()
