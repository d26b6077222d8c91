"""The row-pruned LAST layer of a token-stationary rollout unit (192 videos, C2): ms per 50-step rollout (hipGraph replay, whole chip, alone) under the kernel forms
the per-call options select for it.    python tools/tok_last_layer_probe.py
(the all-heads variants need the engine switch of profiles/r06_probes.txt section 22, which was removed after the measurement: on the committed library every
variant runs the row-tile form.)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine  # noqa: E402
from slotformer_amd.pipeline import pair_unit_options  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
B = 192
base = pair_unit_options(roll, 32, 6, 128, True, 6)
variants = [('pipeline default', base), ('ffn_rows 64', dict(base, ffn_rows=64)),
            ('all-heads last layer (attn_heads 8, attn_rows 0, ffn_tile 0)', dict(base, attn_rows=0, ffn_tile=0, attn_heads=8)),
            ('all-heads last layer, ffn_rows 64', dict(base, attn_rows=0, ffn_tile=0, attn_heads=8, ffn_rows=64))]
x0 = torch.randn(B, 6, 7, 128, device=dev)


def graph_ms(opts, n=5):
    buf = torch.zeros(B, 56, 7, 128, device=dev)
    buf[:, :6] = x0
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
    return (time.perf_counter() - t0) / n * 1e3, buf


with torch.no_grad():
    ref = None
    for name, o in variants:
        try:
            ms, buf = graph_ms(o)
        except Exception as e:  # noqa: BLE001
            print(f'{name:60s} refused: {str(e)[:80]}')
            continue
        if ref is None:
            ref = buf.clone()
        print(f'{name:60s} {ms:7.2f} ms per unit  ({1e3 * ms / 50:6.1f} us per step)   max |diff| vs default {(buf - ref).abs().max().item():.2e}', flush=True)
