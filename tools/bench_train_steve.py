"""Training-step benchmark of STEVE at the reference's Physion training shape (steve_physion_params.py:
12 clips per GPU x 6 frames at 128x128, 6 slots of 192, Transformer + LSTM predictor, frozen dVAE with 4096 tokens on a
32x32 grid, 4-block Transformer decoder of width 192, token cross-entropy, Adam).

  python tools/bench_train_steve.py [--batch 12] [--steps 5] [--warmup 1] [--eager]

One JSON line: ms per iteration of the HIP path (forward + loss + backward + optimiser); --eager adds the same step with torch
ops (autograd) on the same GPU, sharing the parameters and the (inference-path) dVAE tokens.
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
from bench_train_ops import eager_slot_attention  # noqa: E402


def physion_cfg():
    cfg = gu.savi_cfg(128, 6, slot_size=192, mlp=384, iters=2, pred='transformer', rnn=True, kld='none', enc_out=192, pred_ffn=768)
    cfg['model'] = 'STEVE'
    cfg['dvae_dict'] = dict(down_factor=4, vocab_size=4096, dvae_ckp_path='')
    cfg['dec_dict'] = dict(dec_type='slate', dec_num_layers=4, dec_num_heads=4, dec_d_model=192)
    cfg['loss_dict'] = dict(use_img_recon_loss=False)
    return cfg


def eager_decoder(dec, slots, idx):
    """STEVETransformerDecoder.forward (steve_transformer.py:275-303) with torch ops on the module's parameters."""
    B, T = idx.shape
    H, d = dec.n_head, dec.d_model
    mem = dec.in_proj(slots)
    tokens = torch.cat([torch.full((B, 1), dec.vocab_size, dtype=torch.int64, device=idx.device), idx], 1)
    x = dec.pos_emb.dropout(dec.tok_emb(tokens) + dec.pos_emb.pe[:, :T + 1])
    L = T + 1
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool, device=idx.device), 1)

    def mha(m, q, k, v, mask):
        Bq, Tq, _ = q.shape
        S = k.shape[1]
        hd = d // H
        q = m.proj_q(q).view(Bq, Tq, H, hd).transpose(1, 2) * hd**-0.5
        k = m.proj_k(k).view(Bq, S, H, hd).transpose(1, 2)
        v = m.proj_v(v).view(Bq, S, H, hd).transpose(1, 2)
        a = q @ k.transpose(-1, -2)
        if mask is not None:
            a = a.masked_fill(mask, float('-inf'))
        a = m.attn_dropout(a.softmax(-1))
        return m.output_dropout(m.proj_o((a @ v).transpose(1, 2).reshape(Bq, Tq, d)))

    for blk in dec.tf_dec.blocks:
        y = blk.self_attn_layer_norm(x)
        if blk.is_first:
            x = y
        x = x + mha(blk.self_attn, y, y, y, causal)
        y = blk.encoder_decoder_attn_layer_norm(x)
        x = x + mha(blk.encoder_decoder_attn, y, mem, mem, None)
        x = x + blk.ffn(blk.ffn_layer_norm(x))
    return dec.head(dec.tf_dec.layer_norm(x))


def eager_forward(m, img, tgt):
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
    prev, state, posts = None, None, []
    pred = m.predictor
    for t in range(T):
        if prev is None:
            lat = m.init_latents.repeat(B, 1, 1)
        else:
            h = pred.base_predictor.transformer_encoder(prev)
            o, state = pred.rnn(h.reshape(1, -1, h.shape[-1]), state)
            lat = pred.out_projector(o[0]).view(h.shape)
        prev = eager_slot_attention(m.slot_attention, feats[:, t], lat)
        posts.append(prev)
    slots = torch.stack(posts, 1).flatten(0, 1)
    logits = eager_decoder(m.trans_decoder, slots, tgt[:, :-1])
    return F.cross_entropy(logits.flatten(0, 1), tgt.flatten(0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=12)
    ap.add_argument('--frames', type=int, default=6)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--eager', action='store_true')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    from slotformer_amd.base_slots import build_model
    from slotformer_amd import train
    torch.manual_seed(0)
    m = build_model(gu.ParamsView(physion_cfg())).to(dev).train()
    m.testing = False
    B, T = a.batch, a.frames
    img = torch.rand(B, T, 3, 128, 128, device=dev) * 2 - 1
    with torch.no_grad():
        tok = m.dvae.tokenize(img, one_hot=False).flatten(2, 3)           # [B,T,1024] targets from the frozen dVAE
    data = {'img': img, 'token_id': tok}
    opt = train.FlatAdam([p for p in m.parameters() if p.requires_grad], lr=1e-4)

    def step():
        opt.zero_grad()
        out = m(data)
        m.calc_train_loss(data, out)['token_recon_loss'].backward()
        opt.step()

    ms = time_loop(step, a.steps, a.warmup)
    res = {'metric': 'steve_training_iterations_per_sec', 'value': round(1e3 / ms, 2), 'unit': 'it/s', 'ms_per_iter': round(ms, 1),
           'frames_per_sec': round(B * T * 1e3 / ms, 1),
           'config': {'workload': f'STEVE Physion training step, B={B}, T={T}, 128x128, 6 slots of 192, 1025-token decoder (4 blocks, '
                                  'width 192, vocab 4096), all dropouts on, token cross-entropy, Adam', 'dtype': 'f32 (split-bf16 MFMA)'}}
    if a.eager:
        tgt = tok.flatten(0, 1).long()

        def estep():
            opt.zero_grad()
            m.predictor.reset()
            eager_forward(m, img, tgt).backward()
            opt.step()

        ems = time_loop(estep, a.steps, a.warmup)
        res['torch_eager_same_gpu'] = {'ms_per_iter': round(ems, 1), 'speedup': round(ems / ms, 2)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
