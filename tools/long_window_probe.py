"""The reference's Physion window (slotformer_physion_params.py: 15 burn-in frames x 6 slots = 90 tokens, 8 layers, slot size 192) through sf_rollout_f32
with the layers before the last as token-stationary launches (one video per workgroup) against the long-window forms: difference, us per step.

    python tools/long_window_probe.py [B ...]      (default 16 64 96 192)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402
from slotformer_amd import engine  # noqa: E402
from slotformer_amd.video_prediction.models import SlotRollouter  # noqa: E402

dev = torch.device('cuda:0')
rd = dict(gu.C4_ROLL_REF['rollout_dict'])
torch.manual_seed(5)
roll = SlotRollouter(**rd).eval().to(dev)
hist, N, Cs = rd['history_len'], rd['num_slots'], rd['slot_size']
H = 10


def graph_ms(buf, opts, n=5):
    for _ in range(2):
        engine.rollout(roll, buf, hist, H, opts=opts)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        engine.rollout(roll, buf, hist, H, opts=opts)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for B in [int(a) for a in sys.argv[1:]] or [16, 64, 96, 192]:
        x0 = torch.randn(B, hist, N, Cs, device=dev)

        def fresh():
            buf = torch.zeros(B, hist + H, N, Cs, device=dev)
            buf[:, :hist] = x0
            return buf

        a = engine.rollout(roll, fresh(), hist, H, opts={'layer_tok': True}).clone()
        b = engine.rollout(roll, fresh(), hist, H, opts={'layer_tok': False}).clone()
        d = ((a - b).abs().max() / b.abs().max()).item()
        t_tok = graph_ms(fresh(), {'layer_tok': True})
        t_row = graph_ms(fresh(), {'layer_tok': False})
        print(f'B {B:4d} window {hist * N} tokens, {rd["num_layers"]} layers: token-stationary {1e3 * t_tok / H:8.1f} us per step   long-window forms {1e3 * t_row / H:8.1f} us per step   '
              f'max rel difference {d:.2e}', flush=True)
