"""Timing of the weights-stationary 5x5 convolution (csrc/conv_ws.hip) against the 4-row-tile kernel: whole chip and on a 128-CU masked stream, one time
step (32 frames) and six (192 frames) per launch; phase stamps of workgroup 0 with SF_DBG=conv.   python tools/conv_ws_probe.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slotformer_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda:0')
lib = _lib.lib()
torch.manual_seed(0)
w = torch.randn(64, 64, 5, 5, device=dev) * 0.03
b = torch.randn(64, device=dev) * 0.1
wf = ops.pack_conv_frag(ops.pack_conv_weight(w))


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


h = C.c_void_p()
_lib.check(lib.sf_stream_create_cu_mask(C.byref(h), (C.c_uint * 8)(*([0] * 4 + [0xffffffff] * 4)), 8))
masked = torch.cuda.ExternalStream(h.value, device=dev)
for F_ in (32, 192):
    x = torch.randn(F_, 64, 64, 64, device=dev)
    fl = 2.0 * F_ * 4096 * 64 * 1600
    ref = ops.conv5x5_frag(x, wf, b)
    out = ops.conv5x5_ws(x, wf, b)
    print(f'F={F_}: bit-identical {torch.equal(ref, out)}', flush=True)
    for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
        with torch.cuda.stream(st):
            t_t = timeit(lambda: ops.conv5x5_frag(x, wf, b))
            t_w = timeit(lambda: ops.conv5x5_ws(x, wf, b))
            extra = ''
            if name == 'whole chip':
                extra = '  ws with 128 / 512 workgroups: ' + ' / '.join(f'{timeit(lambda: ops.conv5x5_ws(x, wf, b, n_workgroups=n)):.1f}' for n in (128, 512))
        cus = lib.sf_stream_cus(C.c_void_p(st.cuda_stream))
        roof = 2500.0 / 3 * cus / 256
        print(f'  {name:12s} ({cus} CUs): tiles {t_t:7.1f} us ({fl / t_t / 1e6 / roof:.3f} of the roof)   stationary {t_w:7.1f} us ({fl / t_w / 1e6 / roof:.3f})' + extra, flush=True)
if 'conv' in os.environ.get('SF_DBG', ''):
    ts = (C.c_longlong * 16)()
    x = torch.randn(32, 64, 64, 64, device=dev)
    ops.conv5x5_ws(x, wf, b)
    torch.cuda.synchronize()
    lib.sf_debug_read_ts_conv_ws(ts)
    t = list(ts)
    print('stamps (100 MHz ticks -> us): first window', (t[1] - t[0]) / 100, ' weights + rows', (t[2] - t[1]) / 100, ' last epilogue', (t[3] - t[2]) / 100)
    if t[14]:
        n = t[14]
        print(f'  per row of workgroup 0 (shader cycles, {n} rows): first block per wave', [round(v / n) for v in t[4:8]], ' own block', [round(v / n) for v in t[8:12]],
              ' wave 0 at the mid barrier', round(t[12] / n), ' at the end barrier', round(t[13] / n))
