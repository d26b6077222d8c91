"""TWO ranks of the multi-GPU path on the one GPU of the box (process group: gloo -- RCCL refuses two ranks on one device):

* each rank runs `harness.extract_and_rollout` (the product pipeline: encode graphs, rollout units, CU-masked streams) on ITS
  `shard_range` of the videos, rank 0 gathers the shards (`parallel.gather_shards`) and the result equals the single-process run
  over all videos bit for bit -- the path has no data-path collective (the reference: one process per GPU,
  scripts/sbatch_run.sh:36-42, DataParallel in base_slots/extract_slots.py:128);
* `parallel.sharded_extract` (H1 across ranks) the same way;
* each rank runs one real SlotFormer training step (forward + calc_train_loss + backward) on its half of a batch with
  `ddp_flat_bucket=True`: the in-backward all-reduce (sum -> mean over TWO ranks, real gradients) must leave every rank with the
  single-process gradients of the whole batch (scripts/train.py:85,103 -> DDP).
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import golden_util as gu

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _models(dev):
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(11)
    savi = build_model(gu.ParamsView(gu.C2_SAVI)).eval().to(dev)
    savi.testing = True
    roll = SlotRollouter(**gu.C2_ROLL['rollout_dict']).eval().to(dev)
    return savi, roll


def _train_model(dev):
    from test_engine_gpu import build
    cfg = gu.TRAIN_ROLL
    m, _ = build(cfg, gu.load_golden('roll_train'), 801, dev, vp=True)
    m.train()
    for mod in m.modules():   # dropout off: the two halves must see the arithmetic of the whole batch
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.
    rd = cfg['rollout_dict']
    slots = gu.seeded_normal((4, rd['history_len'] + cfg['loss_dict']['rollout_len'], rd['num_slots'], rd['slot_size']), 5).to(dev)
    return m, slots


def _grads(m, slots, flag):
    m.rollouter.ddp_flat_bucket = flag
    m.zero_grad(set_to_none=True)
    out = m({'slots': slots})
    m.calc_train_loss({'slots': slots}, out)['slot_recon_loss'].backward()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


V, T, H = 6, 6, 5


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from slotformer_amd import harness, parallel
        dev = torch.device('cuda:0')
        torch.cuda.set_device(dev)
        savi, roll = _models(dev)
        videos = gu.seeded_img(V, T, 128, seed=77)
        noises = gu.seeded_normal((V, T, 7, 128), 78)
        lo, hi = parallel.shard_range(V, rank, world)
        with torch.no_grad():
            # the rank's share through the pipeline: one video per batch -> three pipelined batches
            local = harness.extract_and_rollout(savi, roll, videos[lo:hi].to(dev), H, batch_size=1, noises=noises[lo:hi])
            full = parallel.gather_shards(local.cpu(), V)
            # H1 across ranks (deterministic model: the PHYRE SAVi samples nothing)
            from slotformer_amd.base_slots import build_model
            torch.manual_seed(12)
            det = build_model(gu.ParamsView(gu.C5_SAVI)).eval().to(dev)
            det.testing = True
            vids1 = gu.seeded_img(5, 2, 128, seed=79)
            sh = parallel.sharded_extract(det, vids1, batch_size=2)
        # one training step on this rank's half of the batch, gradients averaged over the two ranks inside backward()
        m, slots = _train_model(dev)
        b0, b1 = parallel.shard_range(slots.shape[0], rank, world)
        g = _grads(m, slots[b0:b1].contiguous(), True)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            harness.release_pipelines()
            with torch.no_grad():
                ref = harness.extract_and_rollout(savi, roll, videos.to(dev), H, batch_size=2, noises=noises).cpu()
                ref1 = harness.extract_video_slots(det, vids1, batch_size=2)
            gref = _grads(m, slots, False)
            errs = {n: ((g[n] - gref[n]).norm() / gref[n].norm().clamp_min(1e-30)).item() for n in gref}
            q.put({'slots_equal': bool(torch.equal(full, ref)), 'finite': bool(torch.isfinite(full).all()), 'shape': tuple(full.shape),
                   'h1_equal': bool(torch.equal(sh, ref1)), 'grad_keys': sorted(g) == sorted(gref), 'n_grads': len(gref),
                   'grad_err': max(errs.values()), 'grad_err_at': max(errs, key=errs.get),
                   'half_differs': bool(any(not torch.equal(_grads(m, slots[b0:b1].contiguous(), False)[n], gref[n]) for n in list(gref)[:4]))})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu(dev):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = q.get(timeout=900)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()   # (the process this test started, by handle)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert res['shape'] == (V, T + H, 7, 128) and res['finite']
    assert res['slots_equal'], 'the gathered shards of two ranks differ from the single-process result'
    assert res['h1_equal'], 'parallel.sharded_extract differs from the single-process extract_video_slots'
    assert res['grad_keys'] and res['n_grads'] > 10
    assert res['half_differs'], 'the half-batch gradient equals the whole-batch one: the test would not see a missing all-reduce'
    # the two halves are summed in another order than the whole batch: 1e-6 relative (L2) per parameter
    assert res['grad_err'] < 1e-6, (res['grad_err'], res['grad_err_at'])
