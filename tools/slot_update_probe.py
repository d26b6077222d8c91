"""Slot update (savi.py:95-100 + project_q) per launch: the VALU kernel against the matrix-core kernels (slot_update_mfma.hip at slot size 128,
slot_update_wide.hip at 192), weights packed once.
    python tools/slot_update_probe.py [D] [B] [N]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from slotformer_amd import ops  # noqa: E402
from slotformer_amd._lib import lib, check  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 6
H, P = 2 * D, 16
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)  # noqa: E731
pn, pd, prev = r(B, P, N, D), 0.5 + r(B, P, N).abs(), r(B, N, D)
w_ih, w_hh, b_ih, b_hh = r(3 * D, D, sc=D**-0.5), r(3 * D, D, sc=D**-0.5), r(3 * D, sc=0.1), r(3 * D, sc=0.1)
lg, lb, w1, b1, w2, b2 = 1 + r(D, sc=0.1), r(D, sc=0.1), r(H, D, sc=D**-0.5), r(H, sc=0.1), r(D, H, sc=H**-0.5), r(D, sc=0.1)
qg, qb, qw = 1 + r(D, sc=0.1), r(D, sc=0.1), r(D, D, sc=D**-0.5)
packed = [ops.pack_linear(w) for w in (w_ih, w_hh, w1, w2, qw)]
tr = [w.t().contiguous() for w in (w_ih, w_hh, w1, w2)]
out, q = torch.empty_like(prev), torch.empty_like(prev)
p = lambda x: x.data_ptr()  # noqa: E731
st = torch.cuda.current_stream().cuda_stream


def mfma():
    check(lib().sf_slot_update_packed_f32(p(pn), p(pd), P, p(prev), p(packed[0]), p(packed[1]), p(b_ih), p(b_hh), p(lg), p(lb), p(packed[2]), p(b1),
                                          p(packed[3]), p(b2), p(out), p(qg), p(qb), p(packed[4]), p(q), B, N, D, H, 1e-5, st))


def valu():
    check(lib().sf_slot_update_f32(p(pn), p(pd), P, p(prev), p(tr[0]), p(tr[1]), p(b_ih), p(b_hh), p(lg), p(lb), p(tr[2]), p(b1), p(tr[3]), p(b2), p(out),
                                   B, N, D, H, 1e-5, st))


for name, fn in (('valu (no q projection)', valu), ('matrix cores (+ q projection)', mfma)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f'D {D} rows {B * N}: {name:32s} {e0.elapsed_time(e1) * 5:.2f} us per launch')
