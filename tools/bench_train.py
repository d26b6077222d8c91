"""Training-step benchmark of the rollout Transformer at the reference's CLEVRER training shape
(slotformer_clevrer_params.py: 6 burn-in + 10 rollout frames, 7 slots, d_model 256, 4 layers, dropout 0.1, Adam 2e-4,
batch 32 per GPU), slot-reconstruction loss only.

  python tools/bench_train.py [--batch 32] [--steps 20] [--warmup 3] [--eager] [--phases]

Prints one JSON line: iterations/s and ms per iteration of the HIP path (forward + loss + backward + Adam step);
--eager adds the same step written the reference's way (torch autograd over nn.TransformerEncoder calls on the ROCm
PyTorch build, slotformer.py:110-124) on the same GPU.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

import golden_util as gu  # noqa: E402


def build(dev, S, img=False):
    from slotformer_amd.base_slots import build_model as bb
    from slotformer_amd.video_prediction import build_model as bv
    cfg = {**gu.C2_ROLL, 'loss_dict': dict(rollout_len=S, use_img_recon_loss=img)}
    scfg = gu.savi_cfg(64, 7)
    scfg['dec_dict'] = {k: v for k, v in cfg['dec_dict'].items() if k != 'dec_ckp_path'}
    torch.manual_seed(0)
    savi = bb(gu.ParamsView(scfg))
    path = os.path.join(tempfile.mkdtemp(), 'savi.pth')
    torch.save({'state_dict': savi.state_dict()}, path)
    full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    full['dec_dict']['dec_ckp_path'] = path
    return bv(gu.ParamsView(full)).to(dev).train(), cfg


def eager_rollout(r, x, pred_len):
    """The reference's formulation (slotformer.py:85-126) on torch's own ROCm kernels."""
    B, N = x.shape[0], x.shape[2]
    in_x = x.flatten(1, 2)
    pe = r.enc_t_pe.unsqueeze(2).repeat(B, 1, N, 1).flatten(1, 2)
    out = []
    for _ in range(pred_len):
        h = r.transformer_encoder(r.in_proj(in_x) + pe)
        pred = r.out_proj(h[:, -N:])
        out.append(pred)
        in_x = torch.cat([in_x[:, N:], pred], dim=1)
    return torch.stack(out, 1)


def eager_decode(m, slots):
    """StoSAVi.decode (savi.py:504-525) on torch's own kernels, with the model's (frozen) decoder modules."""
    F_, N, D = slots.shape
    r = m.dec_resolution[0]
    x = slots.reshape(F_ * N, D, 1, 1).expand(-1, -1, r, r)
    pe = m.decoder_pos_embedding
    x = x + pe.dense(pe.grid).permute(0, 3, 1, 2)
    out = m.decoder(x).view(F_, N, 4, m.resolution[0], m.resolution[1])
    masks = torch.softmax(out[:, :, 3:], dim=1)
    return (out[:, :, :3] * masks).sum(1)


def time_loop(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--rollout', type=int, default=10)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--eager', action='store_true')
    ap.add_argument('--flat-adam', action='store_true', help='train.FlatAdam (one HIP launch per step) instead of torch.optim.Adam')
    ap.add_argument('--amp', action='store_true', help='AMP-bf16 policy: library precision mode 2 (train.amp_bf16)')
    ap.add_argument('--img', action='store_true', help='add the image term (use_img_recon_loss=True, 64x64 frames)')
    ap.add_argument('--phases', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if a.amp:
        from slotformer_amd import _lib
        _lib.check(_lib.lib().sf_set_precision(2))
    S, B = a.rollout, a.batch
    m, cfg = build(dev, S, a.img)
    slots = (0.5 * gu.seeded_normal((B, 6 + S, 7, 128), 1)).to(dev)
    data = {'slots': slots}
    if a.img:
        data['img'] = torch.rand(B, 6 + S, 3, 64, 64, device=dev) * 2 - 1
    params = [p for p in m.parameters() if p.requires_grad]
    if a.flat_adam:
        from slotformer_amd import train as sf_train
        opt = sf_train.FlatAdam(params, lr=2e-4)
    else:
        opt = torch.optim.Adam(params, lr=2e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = m(data)
        terms = m.calc_train_loss(data, out)
        loss = terms['slot_recon_loss'] + (terms['img_recon_loss'] if a.img else 0.)
        loss.backward()
        opt.step()
        return loss

    ms = time_loop(step, a.steps, a.warmup)
    res = {'metric': 'slotformer_training_iterations_per_sec', 'value': round(1e3 / ms, 2), 'unit': 'it/s', 'ms_per_iter': round(ms, 3),
           'frames_per_sec': round(B * (6 + S) * 1e3 / ms, 1),
           'config': {'workload': f'SlotFormer CLEVRER training step, B={B}, 6+{S} frames, 7 slots, d=256, 4 layers, '
                                  'dropout 0.1, slot loss' + (' + image loss through the frozen SAVi decoder (64x64)' if a.img else '') + ', Adam', 'dtype': 'f32 storage, single-pass bf16 MFMA (AMP policy)' if a.amp else 'f32 (split-bf16 MFMA)'}}
    if a.phases:
        def fwd():
            with torch.no_grad():
                pass
            return m({'slots': slots})
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        opt.zero_grad(set_to_none=True)
        ev[0].record()
        out = m(data)
        ev[1].record()
        terms = m.calc_train_loss(data, out)
        loss = terms['slot_recon_loss'] + (terms['img_recon_loss'] if a.img else 0.)
        loss.backward()
        ev[2].record()
        opt.step()
        ev[3].record()
        torch.cuda.synchronize()
        res['phases_ms'] = {'forward': round(ev[0].elapsed_time(ev[1]), 3), 'loss_backward': round(ev[1].elapsed_time(ev[2]), 3),
                            'adam': round(ev[2].elapsed_time(ev[3]), 3)}
    if a.eager:
        r = m.rollouter

        def estep():
            opt.zero_grad(set_to_none=True)
            pred = eager_rollout(r, slots[:, :6], S)
            loss = ((pred - slots[:, 6:])**2).mean()
            if a.img:
                rec = eager_decode(m, pred.flatten(0, 1)).unflatten(0, (B, S))
                loss = loss + ((rec - data['img'][:, 6:])**2).mean()
            loss.backward()
            opt.step()

        ems = time_loop(estep, a.steps, a.warmup)
        res['torch_eager_same_gpu'] = {'ms_per_iter': round(ems, 3), 'speedup': round(ems / ms, 2)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
