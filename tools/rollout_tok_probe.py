"""The C2 rollout (6 + 50) with the token-stationary layer launches (sf_rollout_opts.layer_tok) against the row-tile forms the pipeline uses today and the
exact-f32 path: max relative differences and ms per rollout / us per step from hipGraph replays.

    python tools/rollout_tok_probe.py [B ...]      (default 32 128 192)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()["C2"])
lib = _lib.lib()
TILES = {'seam': False, 'attn_rows': 128, 'ffn_tile': 2, 'layer_tok': False}
TOK = {'seam': False, 'layer_tok': True}
LAT = {'layer_tok': False}


def graph_ms(buf, opts, n=10):
    for _ in range(2):
        engine.rollout(roll, buf, 6, 50, opts=opts)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        engine.rollout(roll, buf, 6, 50, opts=opts)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for B in [int(a) for a in sys.argv[1:]] or [32, 128, 192]:
        torch.manual_seed(B)
        x0 = torch.randn(B, 6, 7, 128, device=dev)

        def fresh():
            buf = torch.zeros(B, 56, 7, 128, device=dev)
            buf[:, :6] = x0
            return buf

        a = engine.rollout(roll, fresh(), 6, 50, opts=TOK).clone()
        a2 = engine.rollout(roll, fresh(), 6, 50, opts=TOK).clone()
        b = engine.rollout(roll, fresh(), 6, 50, opts=TILES).clone()
        ref = engine.rollout(roll, fresh(), 6, 50, opts={'precision': 'f32', 'layer_tok': False}).clone()
        torch.cuda.synchronize()
        rel = lambda u, v: ((u - v).abs().max() / v.abs().max()).item()  # noqa: E731
        print(f'B {B}: token-stationary vs exact f32 {rel(a, ref):.2e}   row tiles vs exact f32 {rel(b, ref):.2e}   token-stationary vs row tiles {rel(a, b):.2e}   '
              f'deterministic {bool((a == a2).all())}  finite {bool(torch.isfinite(a).all())}')
        buf = fresh()
        t_tok, t_tile, t_lat = graph_ms(buf, TOK), graph_ms(buf, TILES), graph_ms(buf, LAT)
        print(f'      ms per rollout (graph replay): token-stationary {t_tok:.3f} ({t_tok * 20:.1f} us/step)   row tiles {t_tile:.3f} ({t_tile * 20:.1f} us/step)   '
              f'latency forms {t_lat:.3f} ({t_lat * 20:.1f} us/step)', flush=True)
