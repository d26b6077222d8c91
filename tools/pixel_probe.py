"""Per-pixel chain of the encoder (LN(64) -> fc1 -> ReLU -> fc2 -> LN(128), 32 frames x 4096 pixels per launch as the C2 encode launches it):
form 0 (tile at a time, weights through LDS), 1 (weights in registers, 128-pixel tiles, one workgroup per CU), 2 (64-pixel tiles, two
workgroups per CU); whole chip and on the 128-CU encode mask.   python tools/pixel_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import _lib  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline, encode_mask_words  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
torch.manual_seed(0)
M = 32 * 4096
x = torch.randn(M, 64, device=dev)
g0, b0 = 1 + 0.1 * torch.randn(64, device=dev), 0.1 * torch.randn(64, device=dev)
w1, bb1 = 0.15 * torch.randn(128, 64, device=dev), 0.1 * torch.randn(128, device=dev)
w2, bb2 = 0.1 * torch.randn(128, 128, device=dev), 0.1 * torch.randn(128, device=dev)
g1, b1 = 1 + 0.1 * torch.randn(128, device=dev), 0.1 * torch.randn(128, device=dev)
ptrs = [v.data_ptr() for v in (x, g0, b0, w1, bb1, w2, bb2, g1, b1)]
outs = [torch.empty(M, 128, device=dev) for _ in range(3)]


def run(form):
    _lib.check(lib.sf_pixel_feat_f32(*ptrs, outs[form].data_ptr(), M, 1e-5, form, torch.cuda.current_stream().cuda_stream))


with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, 32, 6, 50)
    masked = pipe._masked_stream(encode_mask_words('rows4'))
    for f in range(3):
        run(f)
    torch.cuda.synchronize()
    print('bit-identical 0/1/2:', bool(torch.equal(outs[0], outs[1])), bool(torch.equal(outs[0], outs[2])))
    for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
        for f in range(3):
            with torch.cuda.stream(st):
                for _ in range(3):
                    run(f)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    run(f)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 20
            print(f'{name:12s} form {f}: {1e6 * dt:7.1f} us per launch of {M} pixels  ({M * (64 + 128) * 4 / dt / 1e12:.2f} TB/s of rows)', flush=True)
