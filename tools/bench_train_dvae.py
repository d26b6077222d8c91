"""Training-step benchmark of the dVAE at the reference's Physion shape (dvae_physion_params.py: 32
single frames per GPU at 128x128, 4096-token vocabulary on a 32x32 grid, Gumbel-softmax at tau 1, MSE image loss, Adam).

  python tools/bench_train_dvae.py [--batch 32] [--steps 10] [--warmup 2] [--eager] [--amp]

One JSON line: ms per iteration of the HIP path (forward + loss + backward + optimiser); --eager adds the same step with torch
ops (NCHW conv2d / group_norm / pixel_shuffle / the reference's gumbel_softmax arithmetic under autograd) on the same GPU and
parameters; both draw fresh Gumbel noise every step, as training does.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import golden_util as gu  # noqa: E402
from bench_train import time_loop  # noqa: E402


def eager_forward(m, img, tau):
    def block(x, blk, padding=0, stride=1):
        return F.relu(F.group_norm(F.conv2d(x, blk.m.weight, None, stride=stride, padding=padding), 1, blk.weight, blk.bias))
    e, d = m.encoder, m.decoder
    x = block(img, e[0], stride=4)
    for i in range(1, 7):
        x = block(x, e[i])
    logits = F.conv2d(x, e[7].weight, e[7].bias)
    z_logits = F.log_softmax(logits, 1)
    gumbel = -(torch.empty_like(z_logits).exponential_() + torch.finfo(torch.float32).tiny).log()   # steve_utils.py:30-35
    z = F.softmax((z_logits + gumbel) / tau, 1)
    x = block(z, d[0])
    x = block(x, d[1], padding=1)
    x = block(x, d[2])
    x = block(x, d[3])
    x = F.pixel_shuffle(block(x, d[4]), 2)
    x = block(x, d[6], padding=1)
    x = block(x, d[7])
    x = block(x, d[8])
    x = F.pixel_shuffle(block(x, d[9]), 2)
    return F.conv2d(x, d[11].weight, d[11].bias)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--vocab', type=int, default=4096)
    ap.add_argument('--eager', action='store_true')
    ap.add_argument('--amp', action='store_true')
    a = ap.parse_args()
    from slotformer_amd import train, _lib
    from slotformer_amd.base_slots.models.dVAE import dVAE
    dev = torch.device('cuda:0')
    if a.amp:
        _lib.check(_lib.lib().sf_set_precision(2))
    torch.manual_seed(0)
    m = dVAE(a.vocab).to(dev).train()
    img = gu.seeded_img(a.batch, 1, 128, seed=3)[:, 0].to(dev)
    data = {'img': img, 'gumbel_tau': 1.0}
    opt = train.FlatAdam(m.parameters(), lr=1e-4)

    def step():
        opt.zero_grad()
        loss = m.loss_function(data)['recon_loss']
        loss.backward()
        opt.step()
        return loss

    res = {'workload': f'dVAE training step, {a.batch} frames 128x128, vocab {a.vocab}', 'precision': 'bf16' if a.amp else 'bf16x3'}
    l_hip = float(step().detach())
    res['hip_ms'] = round(time_loop(step, a.steps, a.warmup), 3)
    if a.eager:
        opt2 = torch.optim.Adam(m.parameters(), lr=1e-4)

        def estep():
            opt2.zero_grad(set_to_none=True)
            loss = F.mse_loss(eager_forward(m, img, 1.0), img)
            loss.backward()
            opt2.step()
            return loss

        l_eager = float(estep().detach())
        res['eager_ms'] = round(time_loop(estep, a.steps, a.warmup), 3)
        res['speedup'] = round(res['eager_ms'] / res['hip_ms'], 2)
        res['loss_hip_first'], res['loss_eager_after_one_hip_step'] = l_hip, l_eager
    print(json.dumps(res))


if __name__ == '__main__':
    main()
