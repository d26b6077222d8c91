"""Row-tile attention form (csrc/attn_rows.hip) against the all-heads form: rollout-only timing of one unit under hipGraph replay,
bit comparison, and (SF_DBG=lf=16) the in-kernel phase ticks of workgroup 0 of the two new kernels.

    [SF_DBG=lf=16] python tools/attn_rows_probe.py [videos] [steps]"""
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

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 50
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
torch.manual_seed(1)
x0 = torch.randn(B, 6, 7, 128, device=dev)
FORMS = {'head pairs (default)': {'ffn_rows': 64, 'seam': False},
         'all heads': {'attn_heads': 8, 'ffn_rows': 128, 'seam': False},
         'row tiles': {'attn_rows': 128, 'ffn_rows': 128, 'seam': False},
         'row tiles + FFN tiles': {'attn_rows': 128, 'ffn_tile': True, 'seam': False},
         'all heads + FFN tiles': {'attn_heads': 8, 'ffn_tile': True, 'seam': False},
         'row tiles, FFN + next q|k|v fused': {'attn_rows': 128, 'ffn_tile': 2, 'seam': False}}


def fresh():
    buf = torch.zeros(B, 6 + H, 7, 128, device=dev)
    buf[:, :6] = x0
    return buf


outs = {}
with torch.no_grad():
    for slot, (name, opts) in enumerate(FORMS.items()):
        buf = fresh()
        for _ in range(2):
            engine.rollout(roll, buf, 6, H, ws_slot=slot, opts=opts)
        torch.cuda.synchronize()
        outs[name] = buf[:, 6:].clone()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            engine.rollout(roll, buf, 6, H, ws_slot=slot, opts=opts)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        tg = (time.perf_counter() - t0) / 5
        print(f'{name:34s} B={B}: graph {1e3 * tg:.3f} ms ({1e6 * tg / H:.1f} us/step)', flush=True)
    ref = outs['head pairs (default)']
    for name, o in outs.items():
        print(f'{name:34s} equal to default: {bool(torch.equal(o, ref))}  max abs diff {(o - ref).abs().max().item():.3e}  finite {bool(torch.isfinite(o).all())}')
    if 'lf=16' in os.environ.get('SF_DBG', ''):
        engine.rollout(roll, fresh(), 6, 3, opts=FORMS['row tiles'])
        torch.cuda.synchronize()
        out = (C.c_longlong * 32)()
        lib.sf_debug_read_ts_rows.argtypes = [C.POINTER(C.c_longlong)]
        lib.sf_debug_read_ts_rows(out)
        ts = list(out)
        print('qkv_rows ticks (10 ns):', [t - ts[0] for t in ts[:6]], '(0 entry, 1 gamma/beta, 2 half A planes, 3 half B planes, 4 half A done, 5 end)')
        print('attn_core ticks (10 ns):', [t - ts[8] for t in ts[8:11]], '(entry, core done + O planes, end)')
        engine.rollout(roll, fresh(), 6, 3, opts=FORMS['row tiles + FFN tiles'])
        torch.cuda.synchronize()
        o2 = (C.c_longlong * 16)()
        lib.sf_debug_read_ts_ffn_tile.argtypes = [C.POINTER(C.c_longlong)]
        lib.sf_debug_read_ts_ffn_tile(o2)
        t2 = list(o2)
        print('ffn_tile ticks (10 ns):', [t - t2[0] for t in t2[:6]], '(0 entry, 1 LN planes, 2 hidden planes of chunk 0, 3 chunk 0 done, 4 all chunks, 5 end)')
        engine.rollout(roll, fresh(), 6, 3, opts=FORMS['row tiles, FFN + next q|k|v fused'])
        torch.cuda.synchronize()
        lib.sf_debug_read_ts_ffn_tile(o2)
        t2 = list(o2)
        print('ffn_qkv_tile ticks (10 ns):', [t - t2[0] for t in t2[:10]],
              '(0 entry, 1 LN2 planes, 2 hidden planes of chunk 0, 3 chunk 0 done, 4 all chunks, 5 end, 6 y tile + LN1 planes, 7 / 8 / 9 q / k / v written)')
