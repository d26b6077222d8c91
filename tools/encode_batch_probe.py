"""Does the encode of a batch get cheaper per video when two batches are encoded as one (B = 64)?  Whole chip and on a CU-masked
stream of 128 CUs (the encode partition of the 'pair' pipeline):  python tools/encode_batch_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine, _lib  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline, encode_mask_words  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
lib = _lib.lib()
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, 32, 6, 50)
    masked = pipe._masked_stream(encode_mask_words('rows4'))
    for B in (32, 64, 96):
        img = bench.synthetic_img(B, 6, 128).to(dev)
        noise = torch.randn(B, 6, 7, 128, device=dev)
        for name, st in (('whole chip', torch.cuda.current_stream()), ('128-CU mask', masked)):
            with torch.cuda.stream(st):
                for _ in range(3):
                    engine.savi_encode(savi, img, noise=noise, ws_slot=('probe', B))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    engine.savi_encode(savi, img, noise=noise, ws_slot=('probe', B))
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 10
            print(f'encode B={B:3d} {name:12s}: {1e3 * dt:.3f} ms  = {1e3 * dt * 32 / B:.3f} ms per 32 videos', flush=True)
        lib.sf_profile_enable(0x7f)
        bench.read_profile(lib)
        engine.savi_encode(savi, img, noise=noise, ws_slot=('probe', B))
        torch.cuda.synchronize()
        lib.sf_profile_enable(0)
        for k, v in bench.read_profile(lib).items():
            print(f'     {k:26s} launches {v["launches"]:3d}  avg {v["avg_us"]:8.2f} us  total {v["total_ms"]:.3f} ms')
