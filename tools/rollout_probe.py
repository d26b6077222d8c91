"""Rollout-only timing + phase timestamps + a quick cross-check for the C2 rollouter (B=32, 6+50):

    [SF_DBG=lf=16] python tools/rollout_probe.py [B]

prints ms per 50-step rollout (eager and hipGraph replay), us/step, the max relative difference between the fused
split-bf16 path and the exact-f32 GEMM path on the same input, and (with SF_DBG=lf=16) the in-kernel phase ticks."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
lib = _lib.lib()
torch.manual_seed(1)
x0 = torch.randn(B, 6, 7, 128, device=dev)


def fresh():
    buf = torch.zeros(B, 56, 7, 128, device=dev)
    buf[:, :6] = x0
    return buf


with torch.no_grad():
    a = engine.rollout(roll, fresh(), 6, 50).clone()
    lib.sf_set_precision(0)
    ref = engine.rollout(roll, fresh(), 6, 50).clone()
    lib.sf_set_precision(1)
    a2 = engine.rollout(roll, fresh(), 6, 50).clone()
    torch.cuda.synchronize()
    err = ((a - ref).abs().max() / ref.abs().max()).item()
    print(f'fused bf16x3 vs exact-f32 path over 50 steps: max rel err {err:.3e}; deterministic: {bool((a == a2).all())}')
    buf = fresh()
    for _ in range(2):
        engine.rollout(roll, buf, 6, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        engine.rollout(roll, buf, 6, 50)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / 5
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        engine.rollout(roll, buf, 6, 50)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 10
    print('seam timeouts:', lib.sf_seam_timeouts())
    print(f'B={B}: eager {1e3 * te:.3f} ms ({1e6 * te / 50:.1f} us/step)   graph {1e3 * tg:.3f} ms ({1e6 * tg / 50:.1f} us/step)')
    if 'lf=16' in os.environ.get('SF_DBG', ''):
        engine.rollout(roll, buf, 6, 3)
        torch.cuda.synchronize()
        out = (C.c_longlong * 32)()
        lib.sf_debug_read_ts.argtypes = [C.POINTER(C.c_longlong)]
        lib.sf_debug_read_ts(out)
        ts = list(out)
        print('ffn  ticks (10 ns):', [t - ts[0] for t in ts[:9]])
        print('seam attn ticks vs the seam FFN block 0 entry (10 ns):', [t - ts[0] for t in ts[9:15]])
        print('attn ticks (10 ns):', [t - ts[16] for t in ts[16:30]])
