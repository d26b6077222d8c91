"""Timeline of one pipelined run of the bench workload (device times of every encode end, rollout start / end):
   SF_PIPE_TRACE=1 python tools/pipe_timeline.py [n_batches]     (+ the SF_PIPE_* switches of slotformer_amd/switches.py)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['SF_PIPE_TRACE'] = '1'
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
cfg = bench.bench_configs()['C2']
savi, roll = bench.build_models(dev, cfg)
B, T, H = 32, 6, 50
ring = [bench.synthetic_img(B, T, 128, seed=1234 + 1000 * k).to(dev) for k in range(3)]
steal = None
group = os.environ.get('SF_PIPE_GROUP')
cu = 'ff'
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, B, T, H, steal_steps=None if steal is None else float(steal),
                                 group=None if group is None else int(group), encode_cu_word=cu if cu.startswith('rows') else int(cu, 16))
    imgs = [ring[j % 3] for j in range(n)]
    out = torch.empty(n, B, T + H, 7, 128, device=dev)
    pipe.run(imgs[:6], None, out=out[:6])
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    pipe.run(imgs, None, out=out)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tl = pipe.timeline
    print(f'n={n} wall {1e3 * wall:.2f} ms ({n * B * (T + H) / wall / 1e3:.1f} k frames/s)  group {pipe.G} steal {pipe.steal} ramp {pipe.ramp}')
    print('encode end  :', ' '.join(f'{t:6.2f}' for t in tl['encode_end_ms']))
    for (u0, nb), rs, re_ in zip(tl['units'], tl['rollout_start_ms'], tl['rollout_end_ms']):
        print(f'unit batches {u0:2d}..{u0 + nb - 1:2d}: rollout {rs:6.2f} -> {re_:6.2f}  ({re_ - rs:5.2f} ms; waited {rs - tl["encode_end_ms"][u0 + nb - 1]:5.2f} after its last encode)')
