"""CPU, world_size 2 over gloo: the batch-sharded multi-GPU path (no data-path collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from slotformer_amd.parallel import shard_range, max_over_ranks, gather_shards
    n = 7
    lo, hi = shard_range(n, rank, world)
    # every rank "processes" its videos independently: result = f(video index), no communication
    local = torch.stack([torch.full((3, ), float(i)) for i in range(lo, hi)])
    full = gather_shards(local, n)
    t = max_over_ranks(1.0 + rank)
    # flat-bucket gradient all-reduce (training config C3): rank r holds grads filled with r+1 -> mean 1.5
    from slotformer_amd.parallel import allreduce_flat_bucket
    ps = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    ps[0].grad = torch.full((3, 4), float(rank + 1))
    ps[1].grad = torch.full((5, ), float(rank + 1))
    nbytes = allreduce_flat_bucket(ps)
    assert nbytes == 17 * 4 and ps[2].grad is None
    assert torch.allclose(ps[0].grad, torch.full((3, 4), 1.5)) and torch.allclose(ps[1].grad, torch.full((5, ), 1.5))
    # the training path's bucket is born flat: views of it are the per-parameter gradients
    from slotformer_amd.parallel import allreduce_flat
    from slotformer_amd.train import _split
    flat = torch.arange(17, dtype=torch.float32) * (rank + 1)
    views = _split(flat, ps[:2])
    assert allreduce_flat(flat) == 17 * 4
    assert torch.allclose(flat, torch.arange(17, dtype=torch.float32) * 1.5)
    assert views[0].shape == (3, 4) and torch.allclose(views[1], torch.arange(12, 17, dtype=torch.float32) * 1.5)
    dist.barrier()
    if rank == 0:
        q.put((full.tolist(), t))
    dist.destroy_process_group()


def test_sharded_gather_and_max_time():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert full == [[float(i)] * 3 for i in range(7)]
    assert t == pytest.approx(2.0)
