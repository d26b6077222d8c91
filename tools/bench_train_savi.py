"""Training-step benchmark of StoSAVi at the reference's CLEVRER training shape
(stosavi_clevrer_params.py: 16 clips per GPU x 6 frames at 64x64, 7 slots, 2 Slot-Attention iterations, residual-MLP
predictor, stochastic kernels, loss = post_recon_loss + kld_w * kld_loss, Adam 1e-4).

  python tools/bench_train_savi.py [--batch 16] [--steps 10] [--warmup 2] [--eager]

One JSON line: ms per iteration of the HIP path (forward + loss + backward + Adam); --eager adds the same step written the
reference's way with torch ops (MIOpen convolutions, autograd) on the same GPU, sharing the model's parameters.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import golden_util as gu  # noqa: E402
from bench_train import eager_decode, time_loop  # noqa: E402
from bench_train_ops import eager_slot_attention  # noqa: E402


def eager_forward(m, img, noise):
    """StoSAVi.encode + decode (savi.py:367-416, 504-525) with torch ops on the model's own parameters."""
    B, T = img.shape[:2]
    x = img.flatten(0, 1)
    n = len(m.enc_channels) - 1
    for i in range(n):
        conv = m.encoder[i][0]
        x = F.conv2d(x, conv.weight, conv.bias, stride=conv.stride, padding=conv.padding)
        if i != n - 1:
            x = F.relu(x)
    pe = m.encoder_pos_embedding
    x = x + pe.dense(pe.grid).permute(0, 3, 1, 2)
    feats = m.encoder_out_layer(x.flatten(2, 3).permute(0, 2, 1)).unflatten(0, (B, T))
    D, prev, dists, posts = m.slot_size, None, [], []
    for t in range(T):
        if prev is None:
            lat = m.init_latents.repeat(B, 1, 1)
        else:
            h = m.predictor.ln(prev)
            lat = m.predictor.mlp(h) + h
        dist = m.kernel_dist_layer(lat)
        kernels = dist[..., :D] + noise[:, t] * torch.exp(0.5 * dist[..., D:])
        prev = eager_slot_attention(m.slot_attention, feats[:, t], kernels)
        dists.append(dist)
        posts.append(prev)
    post = torch.stack(posts, 1)
    recon = eager_decode(m, post.flatten(0, 1)).unflatten(0, (B, T))
    return torch.stack(dists, 1), recon


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--frames', type=int, default=6)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--eager', action='store_true')
    ap.add_argument('--flat-adam', action='store_true', help='train.FlatAdam (one HIP launch per step) instead of torch.optim.Adam')
    ap.add_argument('--amp', action='store_true', help='AMP-bf16 policy: library precision mode 2 (train.amp_bf16)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if a.amp:
        from slotformer_amd import _lib
        _lib.check(_lib.lib().sf_set_precision(2))
    from slotformer_amd.base_slots import build_model
    from slotformer_amd.host import losses
    torch.manual_seed(0)
    m = build_model(gu.ParamsView(gu.TRAIN_SAVI)).to(dev).train()
    m.testing = False
    B, T = a.batch, a.frames
    img = torch.rand(B, T, 3, 64, 64, device=dev) * 2 - 1
    noise = torch.randn(B, T, 7, 128, device=dev)
    data = {'img': img, 'noise': noise}
    if a.flat_adam:
        from slotformer_amd import train as sf_train
        opt = sf_train.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    else:
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
    kld_w = 1e-4

    def step():
        opt.zero_grad(set_to_none=True)
        out = m(data)
        terms = m.calc_train_loss(data, out)
        (terms['post_recon_loss'] + kld_w * terms['kld_loss']).backward()
        opt.step()

    ms = time_loop(step, a.steps, a.warmup)
    res = {'metric': 'stosavi_training_iterations_per_sec', 'value': round(1e3 / ms, 2), 'unit': 'it/s', 'ms_per_iter': round(ms, 2),
           'frames_per_sec': round(B * T * 1e3 / ms, 1),
           'config': {'workload': f'StoSAVi CLEVRER training step, B={B}, T={T}, 64x64, 7 slots, 2 SA iterations, MLP predictor, '
                                  'recon + KLD loss, Adam', 'dtype': 'f32 storage, single-pass bf16 MFMA (AMP policy)' if a.amp else 'f32 (split-bf16 MFMA)'}}
    if a.eager:
        def estep():
            opt.zero_grad(set_to_none=True)
            dist, recon = eager_forward(m, img, noise)
            (((recon - img)**2).mean() + kld_w * losses.kernel_kld(dist, m.slot_size, m.kld_log_var)).backward()
            opt.step()

        ems = time_loop(estep, a.steps, a.warmup)
        res['torch_eager_same_gpu'] = {'ms_per_iter': round(ems, 2), 'speedup': round(ems / ms, 2)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
