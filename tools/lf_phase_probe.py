"""Phase timestamps (100 MHz wall clock) of workgroup 0 of the fused rollout-layer kernels:
   SF_LF_DBG=16 python tools/lf_phase_probe.py"""
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
savi, roll = bench.build_models(dev)
lib = _lib.lib()
buf = torch.randn(32, 56, 7, 128, device=dev)
with torch.no_grad():
    for _ in range(2):
        engine.rollout(roll, buf, 6, 3)
    torch.cuda.synchronize()
    out = (C.c_longlong * 32)()
    lib.sf_debug_read_ts.argtypes = [C.POINTER(C.c_longlong)]
    lib.sf_debug_read_ts(out)
    ts = list(out)
    print('ffn  ticks (10 ns):', [t - ts[0] for t in ts[:9]])
    print('attn ticks (10 ns):', [t - ts[16] for t in ts[16:28]])
