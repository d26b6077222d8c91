"""Slot-Attention iteration on the folded features (keys == values, slot size 128): microseconds per launch of 32 frames, whole chip and
on the 128-CU encode mask.  SF_SA_TILE=0 selects the two-pass kernel.   python tools/sa_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops, _lib  # noqa: E402
import ctypes as C  # noqa: E402

dev = torch.device('cuda:0')
B, HW, D, N = int(os.environ.get('B', 32)), 4096, 128, 7
x = torch.randn(B, HW, D, device=dev)
q = torch.randn(B, N, D, device=dev)
lib = _lib.lib()
words = (C.c_uint * 8)(*([0] * 4 + [0xffffffff] * 4))
h = C.c_void_p()
_lib.check(lib.sf_stream_create_cu_mask(C.byref(h), words, 8))
masked = torch.cuda.ExternalStream(h.value, device=dev)
for name, st in (('whole chip', torch.cuda.Stream(device=dev)), ('128-CU mask', masked)):
    with torch.cuda.stream(st):
        for _ in range(5):
            ops.slot_attn_iter(x, x, q)
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            ops.slot_attn_iter(x, x, q)
        st.synchronize()
        dt = (time.perf_counter() - t0) / 50
    print(f'SF_SA_TILE={os.environ.get("SF_SA_TILE", "0")} {name:12s}: {1e6 * dt:6.1f} us per launch of {B} frames  ({B * HW * D * 4 / dt / 1e12:.2f} TB/s of unique bytes)', flush=True)

if 'sa' in os.environ.get('SF_DBG', ''):
    lib.sf_debug_sa_stamps(1)
    for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
        with torch.cuda.stream(st):
            ops.slot_attn_iter(x, x, q)
            st.synchronize()
        o = (C.c_longlong * 16)()
        lib.sf_debug_read_ts_sa(o)
        ts = list(o)
        print(name, 'sa_attn_tile_kernel ticks (10 ns), workgroup 0 wave 0:', [v - ts[0] for v in ts[:8]], '(0 entry, 1 queries staged + barrier, 2-5 tiles 0-3 done, 6 reduction summed, 7 end)')
