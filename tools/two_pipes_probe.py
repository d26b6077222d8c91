"""Two EncodeRolloutPipeline objects in one process (the harness keeps one per shape): what the second one's CU-masked queues cost the first.
    python tools/two_pipes_probe.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

dev = torch.device('cuda:0')
cfg = bench.bench_configs()['C2']
savi, roll = bench.build_models(dev, cfg)[:2]
B, T, H = cfg[3], cfg[4], cfg[5]
n = 20
imgs = [torch.rand(B, T, 3, 128, 128, device=dev) * 2 - 1 for _ in range(3)]
batches = [imgs[j % 3] for j in range(n)]


def timed(pipe, tag):
    for _ in range(2):
        pipe.run(batches)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        pipe.run(batches)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    print(f'{tag:46s} {1e3 * dt:7.1f} ms  {n * B * (T + H) / dt / 1e3:7.1f} k frames/s', flush=True)


with torch.no_grad():
    p1 = EncodeRolloutPipeline(savi, roll, B, T, H)
    timed(p1, 'first pipeline, alone')
    p2 = EncodeRolloutPipeline(savi, roll, B, T, H)
    timed(p2, 'second pipeline, the first one alive')
    timed(p1, 'first pipeline, the second one alive')
    p1.close()
    timed(p2, 'second pipeline, the first one closed')
    p3 = EncodeRolloutPipeline(savi, roll, B, T, H)
    timed(p3, 'third pipeline, first closed, second alive')
