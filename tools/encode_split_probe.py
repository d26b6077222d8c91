"""Would TWO encode streams on the encode partition pay?  Stream F computes the CNN features of all time steps of batch j + 1
(engine.savi_cnn) while stream S runs the per-pixel chain + slot chain of batch j on precomputed features
(engine.savi_encode(feat_pre=...)); both CU-masked to the encode partition; optionally beside two rollout graphs replaying on the
rollout partition.   python tools/encode_split_probe.py"""
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
savi, roll = bench.build_models(dev, bench.bench_configs()['C2'])
B, T = 32, 6
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, B, T, 50)
    words = encode_mask_words('rows4')
    sF, sS = pipe._masked_stream(words), pipe._masked_stream(words)
    imgs = [bench.synthetic_img(B, T, 128, seed=k).to(dev) for k in range(3)]
    noise = torch.randn(B, T, 7, 128, device=dev)
    feats = [torch.empty(T, B, 4096, 64, device=dev) for _ in range(2)]
    NB = 12

    def rollouts(on):
        if not on:
            return
        for k, st in enumerate(pipe.roll_streams):
            with torch.cuda.stream(st):
                for _ in range(3):
                    pipe.units[k].graph.replay()

    def single(with_roll):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rollouts(with_roll)
        with torch.cuda.stream(sS):
            for j in range(NB):
                engine.savi_encode(savi, imgs[j % 3], noise=noise, ws_slot=('p', 0))
            e = torch.cuda.Event(); e.record(sS)
        e.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * dt / NB

    def split(with_roll):
        torch.cuda.synchronize()
        evF = [torch.cuda.Event() for _ in range(NB)]
        evS = [torch.cuda.Event() for _ in range(NB)]
        t0 = time.perf_counter()
        rollouts(with_roll)
        for j in range(NB):
            with torch.cuda.stream(sF):
                if j >= 2:
                    sF.wait_event(evS[j - 2])
                engine.savi_cnn(savi, imgs[j % 3], 0, T, out=feats[j & 1], ws_slot=('pf', 0))
                evF[j].record(sF)
            with torch.cuda.stream(sS):
                sS.wait_event(evF[j])
                engine.savi_encode(savi, imgs[j % 3], noise=noise, ws_slot=('ps', 0), feat_pre=feats[j & 1])
                evS[j].record(sS)
        evS[-1].synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * dt / NB

    for fn in (single, split):
        for wr in (False, True):
            fn(wr)
            print(f'{fn.__name__:7s} beside rollouts={wr}: {fn(wr):.3f} ms per batch', flush=True)
    # correctness of the split path
    a = engine.savi_encode(savi, imgs[0], noise=noise, ws_slot=('p', 0))[0]
    f = engine.savi_cnn(savi, imgs[0], 0, T, ws_slot=('pf', 0))
    b = engine.savi_encode(savi, imgs[0], noise=noise, ws_slot=('ps', 0), feat_pre=f)[0]
    torch.cuda.synchronize()
    print('split == single:', bool(torch.equal(a, b)))
