"""5x5 convolution kernels of the encoder (64 -> 64 channels, 64 x 64 grid, 32 frames per launch as the C2 encode launches them):
the 2-row tile kernel (weights through LDS) against the 4-row tile kernel with streamed weight fragments; whole chip and on the
128-CU encode mask.   python tools/conv_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import ops  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline, encode_mask_words  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
torch.manual_seed(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(F, 64, 64, 64, device=dev)
w = torch.randn(64, 64, 5, 5, device=dev) * 0.03
b = torch.randn(64, device=dev) * 0.1
wp = ops.pack_conv_weight(w)
wf = ops.pack_conv_frag(wp)
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, 32, 6, 50)
    masked = pipe._masked_stream(encode_mask_words('rows4'))
    a, c = ops.conv2d_nhwc(x, wp, b), ops.conv5x5_frag(x, wf, b)
    print('bit-identical:', bool(torch.equal(a, c)), ' max diff', (a - c).abs().max().item())
    for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
        for kname, fn in (('2-row tiles, weights via LDS', lambda: ops.conv2d_nhwc(x, wp, b)), ('4-row tiles, streamed fragments', lambda: ops.conv5x5_frag(x, wf, b))):
            with torch.cuda.stream(st):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 20
            fl = 2.0 * F * 4096 * 64 * 64 * 25
            print(f'{name:12s} {kname:32s}: {1e6 * dt:7.1f} us per launch of {F} frames  ({fl / dt / 1e12:.0f} TFLOP/s)', flush=True)
    if ('conv' in os.environ.get('SF_DBG', '')):
        import ctypes as C
        from slotformer_amd import _lib
        lib = _lib.lib()
        ops.conv5x5_frag(x, wf, b)
        torch.cuda.synchronize()
        o = (C.c_longlong * 16)()
        lib.sf_debug_read_ts_conv.argtypes = [C.POINTER(C.c_longlong)]
        lib.sf_debug_read_ts_conv(o)
        ts = list(o)
        print('conv_rows4 ticks (10 ns):', [v - ts[0] for v in ts[:8]], '(0 entry, 1 halo planes written, 2 barrier, 3 after tap 4, 4 after tap 14, 5 taps done, 6 barrier, 7 end)')
        print('taps done per wave:', [v - ts[0] for v in ts[8:16]])
