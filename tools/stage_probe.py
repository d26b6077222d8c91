"""Per-stage start / end times inside the real pipeline schedule:  PART=pair|three python tools/stage_probe.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from slotformer_amd.pipeline import EncodeRolloutPipeline  # noqa: E402

dev = torch.device('cuda:0')
savi, roll = bench.build_models(dev)
ring = [bench.synthetic_img(32, seed=100 + k).to(dev) for k in range(3)]
with torch.no_grad():
    pipe = EncodeRolloutPipeline(savi, roll, 32, 6, 50, partition=os.environ.get('PART', 'pair'),
                                steal_steps=int(os.environ['STEAL']) if 'STEAL' in os.environ else None)
    n = 16
    out = torch.empty(n, 32, 56, 7, 128, device=dev)
    pipe.run([ring[j % 3] for j in range(n)], None, out=out)
    recs = []
    enc0, roll0 = pipe._encode, pipe._rollout

    def enc(*a, **k):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); enc0(*a, **k); e1.record(st)
        recs.append(('E%d' % (a[6] if len(a) > 6 else 9), e0, e1))

    def rol(gi):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); roll0(gi); e1.record(st)
        recs.append(('R', e0, e1))

    pipe._encode, pipe._rollout = enc, rol
    pipe.run([ring[j % 3] for j in range(n)], None, out=out)
    torch.cuda.synchronize()
    base = recs[0][1]
    for name, e0, e1 in recs:
        print(f'  {name}: start {base.elapsed_time(e0):8.2f}  end {base.elapsed_time(e1):8.2f}  ({e0.elapsed_time(e1):.2f} ms)')
