"""CPU emulation of the arithmetic modes of the rollout's linear layers over the full 6+50 horizon (C2, the reference's
weights/inputs of the roll_c2 fixture seeds): every F.linear of the oracle is replaced by one whose operands are rounded as
the MFMA path would see them.  Answers VERDICT r01 item 1's question: what does single-pass bf16 cost on the 6+50 path?

    python tools/precision_probe.py
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import oracle  # noqa: E402
from oracle import slotformer_oracle as so  # noqa: E402
import golden_util as gu  # noqa: E402
from slotformer_amd.video_prediction.models import SlotRollouter  # noqa: E402

real_linear = F.linear


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def make_linear(mode):
    def lin(x, w, b=None):
        if mode == 'bf16':            # single pass: both operands rounded to 8 mantissa bits
            y = real_linear(bf(x).double(), bf(w).double()).float()
        elif mode == 'bf16x3':        # hi*hi + hi*lo + lo*hi
            xh, wh = bf(x), bf(w)
            xl, wl = bf(x - xh), bf(w - wh)
            y = (real_linear(xh.double(), wh.double()) + real_linear(xh.double(), wl.double()) + real_linear(xl.double(), wh.double())).float()
        elif mode == 'bf16_w2':       # weights hi+lo, activations single bf16 (2 passes)
            xh, wh = bf(x), bf(w)
            wl = bf(w - wh)
            y = (real_linear(xh.double(), wh.double()) + real_linear(xh.double(), wl.double())).float()
        elif mode == 'fp16':          # single pass fp16 operands (11 mantissa bits)
            y = real_linear(x.half().double(), w.half().double()).float()
        else:
            y = real_linear(x, w)
        return y if b is None else y + b
    return lin


rd = gu.C2_ROLL['rollout_dict']
torch.manual_seed(0)
roll = SlotRollouter(**rd).eval()
sd = {'rollouter.' + k: v.detach().clone() for k, v in roll.state_dict().items()}
x = gu.seeded_normal((4, 6, 7, 128), 3)
with torch.no_grad():
    ref = so.rollouter_forward(x.double(), 50, {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}, rd).float()
    for mode in ('f32', 'bf16x3', 'bf16_w2', 'fp16', 'bf16'):
        so.F.linear = make_linear(mode)
        out = so.rollouter_forward(x, 50, sd, rd)
        so.F.linear = real_linear
        errs = [((out[:, s] - ref[:, s]).abs().max() / ref[:, s].abs().max()).item() for s in (0, 9, 24, 49)]
        print(f'{mode:8s} max rel err vs float64 after 1 / 10 / 25 / 50 steps: ' + ' / '.join(f'{e:.2e}' for e in errs))
