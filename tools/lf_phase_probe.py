"""Phase timestamps (100 MHz wall clock) of workgroup 0 of the fused rollout-layer kernels:
   SF_DBG=lf=16 python tools/lf_phase_probe.py [B] [ffn_rows]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 32
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
buf = torch.randn(B, 56, 7, 128, device=dev)
with torch.no_grad():
    for _ in range(2):
        engine.rollout(roll, buf, 6, 3, opts={'ffn_rows': rows, 'seam': False, 'attn_heads': int(os.environ.get('ATTN', '2'))})
    torch.cuda.synchronize()
    out = (C.c_longlong * 64)()
    lib.sf_debug_read_ts.argtypes = [C.POINTER(C.c_longlong)]
    lib.sf_debug_read_ts(out)
    ts = list(out)
    print(f'B={B} ffn_rows={rows}')
    print('ffn_body ticks (10 ns):', [t - ts[0] for t in ts[:9]])
    print('wide ffn ticks (10 ns):', [t - ts[32] for t in ts[32:42]], ' FFN1 half 0 after k-steps 3 / 7 / 11:', [t - ts[32] for t in ts[42:45]])
    print('   (wide kernel: 0 entry, 1 first quarter requested, 2 prologue done, 3/4 FFN1 half 0/1, 5 hidden planes complete, 6/7 FFN2 half 0/1, 8 output tile, 9 end;')
    print('    ffn_body: 0 entry, 1 requests, 2 LN planes, 3 FFN1, 4 hidden, 5 FFN2, 6 partial stored, 7 counter, 8 end)')
    print('attn ticks (10 ns):', [t - ts[16] for t in ts[16:28]])
    print('attn_all ticks (10 ns):', [t - ts[48] for t in ts[48:58]], '(0 entry, 1 requests, 2 LN planes; head pair 1: 3 start, 4 proj, 5 planes, 6 core, 7 O planes, 8 out-proj; 9 end)')
