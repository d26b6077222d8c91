"""The decoder's fragment-weight transposed convolution (deconv_s2.hip): microseconds per launch by input width / image count, and the
in-kernel phase stamps of workgroup 0 (SF_DBG=deconv).   python tools/deconv_probe.py [images]"""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 224
w = torch.randn(64, 64, 5, 5, device=dev) * 0.03
b = torch.randn(64, device=dev) * 0.1
hw, hb = torch.randn(4, 64, device=dev) * 0.2, torch.randn(4, device=dev) * 0.1
frag = ops.pack_deconv_frag(ops.pack_deconv_weight(w))
lib = _lib.lib()


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    for W in (64, 32, 16):
        for r in sorted({R, 126, 252, 256, 512}):
            x = torch.randn(r, W, W, 64, device=dev)
            fl = 2.0 * r * W * W * 64 * 64 * 25
            dt = timed(lambda: ops.deconv5x5s2_frag(x, frag, b))
            wgs = r * W * W // 256
            line = f'W {W:3d} images {r:4d} ({wgs:5d} workgroups = {wgs / 256:5.2f} rounds): {1e6 * dt:7.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s = {fl / dt / 1e12 / 833.3:.3f}'
            if W == 64:
                dth = timed(lambda: ops.deconv5x5s2_head(x, frag, b, hw, hb))
                line += f'   | head form {1e6 * dth:7.1f} us  {fl / dth / 1e12:6.1f} TFLOP/s = {fl / dth / 1e12 / 833.3:.3f}'
            print(line, flush=True)
    if ('deconv' in os.environ.get('SF_DBG', '')):
        lib.sf_debug_read_ts_deconv.argtypes = [C.POINTER(C.c_longlong)]
        for name, fn in (('plain W 64', lambda: ops.deconv5x5s2_frag(torch.randn(R, 64, 64, 64, device=dev), frag, b)),
                         ('head  W 64', lambda: ops.deconv5x5s2_head(torch.randn(R, 64, 64, 64, device=dev), frag, b, hw, hb))):
            fn()
            torch.cuda.synchronize()
            o = (C.c_longlong * 16)()
            lib.sf_debug_read_ts_deconv(o)
            ts = list(o)
            print(name, 'ticks (10 ns):', [v - ts[0] for v in ts[:8]], '(0 entry, 1 halo planes written, 2 barrier, 3 after step 11, 4 after step 23, 5 steps done, 7 end)')
            print('   steps done per wave:', [v - ts[0] for v in ts[8:16]])
