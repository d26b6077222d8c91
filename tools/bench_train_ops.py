"""Device time of the training-side kernels outside the rollout (row N1): Slot-Attention iteration backward at the CLEVRER
encode shape (32 videos x 6 frames, 4096 pixels, 7 slots, D = 128) against its HBM roofline (read K, V; write dK, dV)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    B, HW, N, D = 192, 4096, 7, 128
    g = torch.Generator(device='cpu').manual_seed(0)
    k, v = (torch.randn(B, HW, D, generator=g).to(dev) for _ in range(2))
    q, du = (torch.randn(B, N, D, generator=g).to(dev) for _ in range(2))
    pn, pd, _ = ops.slot_attn_iter(k, v, q)
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    fwd = timeit(lambda: ops.slot_attn_iter(k, v, q))
    bwd = timeit(lambda: ops.slot_attn_iter_bwd(k, v, q, pn, pd, du))
    acc = timeit(lambda: ops.slot_attn_iter_bwd(k, v, q, pn, pd, du, dk=dk, dv=dv))
    kv = 2 * B * HW * D * 4
    print(json.dumps({'shape': dict(frames=B, HW=HW, N=N, D=D),
                      'sa_iter_fwd': {'ms': round(fwd, 4), 'GBps': round(kv / fwd / 1e6, 1), 'bytes': kv},
                      'sa_iter_bwd': {'ms': round(bwd, 4), 'GBps': round(2 * kv / bwd / 1e6, 1), 'bytes': 2 * kv},
                      'sa_iter_bwd_accumulate': {'ms': round(acc, 4), 'GBps': round(3 * kv / acc / 1e6, 1), 'bytes': 3 * kv},
                      'hbm_peak_GBps': 8000}))


if __name__ == '__main__':
    main()
