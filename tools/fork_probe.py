"""The encode as two branches (engine.savi_encode side_stream=): ms per C2 batch, eager and from a hipGraph, whole chip and on the 128-CU encode mask.
    python tools/fork_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import engine  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline, encode_mask_words  # noqa: E402

dev = torch.device('cuda:0')
cfg = bench.bench_configs()[os.environ.get('CFG', 'C2')]
savi, roll = bench.build_models(dev, cfg)
B, T = cfg[3], cfg[4]
img = bench.synthetic_img(B, T, 128).to(dev)
noise = engine.kernel_noise(savi, None, B, T, dev)


def timed(fn, st, n=10):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        st.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, B, T, cfg[5])
    masked = pipe._masked_stream(encode_mask_words('rows4'))
    side = torch.cuda.Stream(device=dev)
    ref = engine.savi_encode(savi, img, noise=noise, ws_slot='p0', side_stream=None)[0].clone()
    torch.cuda.synchronize()
    out = engine.savi_encode(savi, img, noise=noise, ws_slot='p1', side_stream=side)[0]
    torch.cuda.synchronize()
    print('forked == plain:', bool(torch.equal(out, ref)))
    for name, st in (('whole chip', torch.cuda.Stream(device=dev)), ('128-CU mask', masked)):
        t_plain = timed(lambda: engine.savi_encode(savi, img, noise=noise, ws_slot='p0', side_stream=None), st)
        t_fork = timed(lambda: engine.savi_encode(savi, img, noise=noise, ws_slot='p1', side_stream=side), st)
        print(f'{name:12s} eager: plain {t_plain:.3f} ms   forked (side = unmasked stream) {t_fork:.3f} ms', flush=True)
        graphs = {}
        for key, sd in (('plain', None), ('fork', torch.cuda.Stream(device=dev))):
            cap = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(cap):
                engine.savi_encode(savi, img, noise=noise, ws_slot=('g', key), side_stream=sd)
                cap.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
                    post = engine.savi_encode(savi, img, noise=noise, ws_slot=('g', key), side_stream=sd)[0]
            torch.cuda.synchronize()
            graphs[key] = (g, post)
        tg = {k: timed(g.replay, st) for k, (g, _) in graphs.items()}
        same = all(bool(torch.equal(p, ref)) for _, p in graphs.values())
        print(f'{name:12s} graph: plain {tg["plain"]:.3f} ms   forked {tg["fork"]:.3f} ms   results equal: {same}', flush=True)
