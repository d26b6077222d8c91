"""Time K/V-cached greedy generation (sf_slate_generate_f32) at the Physion decoder sizes vs steps and batch."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import golden_util as gu
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench_steve_decoder as bsd
from slotformer_amd.base_slots import build_model
dev = torch.device('cuda:0')
with torch.no_grad():
    m = build_model(gu.ParamsView(bsd.cfg_c4())).eval().to(dev)
    for F_ in (12, 1):
        slots = gu.seeded_normal((F_, 6, 192), 4).to(dev)
        for steps in (64, 256, 1024):
            m.trans_decoder.generate_cached(slots, steps)
            torch.cuda.synchronize(); t = time.perf_counter()
            m.trans_decoder.generate_cached(slots, steps)
            torch.cuda.synchronize(); print(F_, steps, round(time.perf_counter() - t, 4), 's', round(1e6 * (time.perf_counter() - t) / steps, 1), 'us/step')
