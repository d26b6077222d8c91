"""The shader clock the chip sustains while the bench pipeline runs (DVFS): a one-wave kernel counts shader cycles (s_memtime) against the constant
100 MHz counter on an idle high-priority stream, launched back to back from a second host thread while the main thread runs the pipelined C2 workload.
Also: the clock with the chip idle and under one rollout alone.      python tools/clock_probe.py [n_batches]"""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd import _lib, engine  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device('cuda:0')
lib = _lib.lib()
cfg = bench.bench_configs()['C2']
savi, roll = bench.build_models(dev, cfg)
B, T, H = 32, 6, 50
ring = [bench.synthetic_img(B, T, 128, seed=1234 + 1000 * k).to(dev) for k in range(3)]
NP = 400
probe_out = torch.zeros(NP, 2, dtype=torch.int64, device=dev)
pstream = torch.cuda.Stream(priority=-1)


def probes(stop, used, us=100):
    """launch clock probes back to back until `stop` is set"""
    i = 0
    torch.cuda.set_device(0)
    while not stop.is_set() and i < NP:
        _lib.check(lib.sf_debug_clock_probe(us, probe_out[i].data_ptr(), pstream.cuda_stream))
        i += 1
        pstream.synchronize()
    used.append(i)


def report(tag, k):
    v = probe_out[:k].cpu().double()
    ghz = v[:, 0] / (v[:, 1] * 10.0)
    if k:
        print(f'{tag}: {k} probes of {v[:, 1].mean().item() / 100:.0f} us: shader clock mean {ghz.mean().item():.3f} GHz  min {ghz.min().item():.3f}  max {ghz.max().item():.3f}', flush=True)


with torch.no_grad():
    # idle chip
    stop, used = threading.Event(), []
    th = threading.Thread(target=probes, args=(stop, used))
    th.start()
    time.sleep(0.02)
    stop.set()
    th.join()
    report('idle chip', used[0])
    # one rollout unit alone (128 videos, row-tile forms)
    buf = torch.randn(128, 56, 7, 128, device=dev)
    opts = {'seam': False, 'attn_rows': 128, 'ffn_tile': 2}
    engine.rollout(roll, buf, 6, 50, opts=opts)
    torch.cuda.synchronize()
    stop, used = threading.Event(), []
    th = threading.Thread(target=probes, args=(stop, used))
    th.start()
    for _ in range(3):
        engine.rollout(roll, buf, 6, 50, opts=opts)
    torch.cuda.synchronize()
    stop.set()
    th.join()
    report('one rollout unit alone (a third of the chip)', used[0])
    # the encode alone on the whole chip
    imgs1 = ring[0]
    engine.savi_encode(savi, imgs1)
    torch.cuda.synchronize()
    stop, used = threading.Event(), []
    th = threading.Thread(target=probes, args=(stop, used))
    th.start()
    for _ in range(10):
        engine.savi_encode(savi, imgs1)
    torch.cuda.synchronize()
    stop.set()
    th.join()
    report('encode alone (whole chip)', used[0])
    # the pipeline
    pipe = EncodeRolloutPipeline(savi, roll, B, T, H)
    imgs = [ring[j % 3] for j in range(n)]
    out = torch.empty(n, B, T + H, 7, 128, device=dev)
    pipe.run(imgs[:8], None, out=out[:8])
    pipe.run(imgs, None, out=out)
    torch.cuda.synchronize()
    stop, used = threading.Event(), []
    th = threading.Thread(target=probes, args=(stop, used))
    t0 = time.perf_counter()
    th.start()
    pipe.run(imgs, None, out=out)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    stop.set()
    th.join()
    print(f'pipeline: {n} batches in {1e3 * wall:.1f} ms ({n * B * (T + H) / wall / 1e3:.1f} k frames/s with the probes running)')
    report('inside the pipelined run (whole chip busy)', used[0])
