"""Whole training step (forward + loss + backward + Adam) of StoSAVi / SlotFormer captured into ONE device graph (torch.cuda.CUDAGraph = hipGraph on ROCm)
against the eager step: ms per iteration and whether the parameters after K steps agree.
    python tools/graph_train_probe.py [savi|slotformer] [--steps 20]"""
import argparse
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import golden_util as gu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('which', nargs='?', default='savi')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
dev = torch.device('cuda:0')
from slotformer_amd.base_slots import build_model  # noqa: E402

torch.manual_seed(0)
m = build_model(gu.ParamsView(gu.TRAIN_SAVI)).to(dev).train()
m.testing = False
B, T = a.batch, 6
img = torch.rand(B, T, 3, 64, 64, device=dev) * 2 - 1
noise = torch.randn(B, T, 7, 128, device=dev)
data = {'img': img, 'noise': noise}
state0 = copy.deepcopy(m.state_dict())
kld_w = 1e-4


def make(capturable):
    m.load_state_dict(state0)
    return torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, capturable=capturable)


def step(opt, set_none=True):
    opt.zero_grad(set_to_none=set_none)
    out = m(data)
    terms = m.calc_train_loss(data, out)
    loss = terms['post_recon_loss'] + kld_w * terms['kld_loss']
    loss.backward()
    opt.step()
    return loss


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# eager
opt = make(False)
for _ in range(3):
    step(opt)
ms_e = timeit(lambda: step(opt), a.steps)
p_e = [p.detach().clone() for p in m.parameters()]
n_e = 3 + a.steps

# graphed
opt = make(True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step(opt)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    static_loss = step(opt)
torch.cuda.synchronize()
n_g = 4   # three warm-up steps + the captured one (capture runs nothing... on ROCm the capture does not execute)
ms_g = timeit(g.replay, a.steps)
print(f'eager {ms_e:.2f} ms  graph {ms_g:.2f} ms  loss {float(static_loss):.5f}')
