"""Encode-only timing with the per-class event breakdown (C2, B=32, 6 frames): python tools/encode_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
lib = _lib.lib()
img = bench.synthetic_img(32).to(dev)
noise = torch.randn(32, 6, 7, 128, device=dev)
with torch.no_grad():
    for _ in range(3):
        engine.savi_encode(savi, img, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        engine.savi_encode(savi, img, noise=noise)
    torch.cuda.synchronize()
    print(f'encode B=32 x 6 frames: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms')
    lib.sf_profile_enable(0x7f)
    bench.read_profile(lib)
    engine.savi_encode(savi, img, noise=noise)
    torch.cuda.synchronize()
    lib.sf_profile_enable(0)
    for k, v in bench.read_profile(lib).items():
        print(f'  {k:26s} launches {v["launches"]:3d}  avg {v["avg_us"]:8.2f} us  total {v["total_ms"]:.3f} ms')
