"""Measure the STEVE image side (row N2) at the reference's Physion sizes (steve_physion_params.py: 128x128 frames,
32x32 = 1024 tokens, vocab 4096, d_model 192, 4 heads, 4 decoder layers, 6 slots): dVAE tokenisation, teacher-forced
Transformer-decoder logits + token cross-entropy, dVAE detokenisation -- HIP path vs the CPU oracle on a bounded sample.
    python tools/bench_steve_decoder.py [--frames 12] [--cpu-frames 2]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

import golden_util as gu  # noqa: E402


def cfg_c4():
    cfg = gu.savi_cfg(128, 6, slot_size=192, mlp=384, iters=2, pred='transformer', rnn=True, kld='none', enc_out=192,
                      pred_layers=2, pred_heads=4, pred_ffn=768)
    cfg['model'] = 'STEVE'
    cfg['dvae_dict'] = dict(down_factor=4, vocab_size=4096, dvae_ckp_path='')
    cfg['dec_dict'] = dict(dec_type='slate', dec_num_layers=4, dec_num_heads=4, dec_d_model=192)
    cfg['loss_dict'] = dict(use_img_recon_loss=False)
    return cfg


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=12)
    ap.add_argument('--cpu-frames', type=int, default=2)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    from slotformer_amd.base_slots import build_model
    import oracle
    dev = torch.device('cuda:0')
    cfg = cfg_c4()
    torch.manual_seed(0)
    m = build_model(gu.ParamsView(cfg)).eval()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev)
    F_ = a.frames
    img = gu.seeded_img(1, F_, 128, seed=3)[0].to(dev)                       # [F,3,128,128]
    slots = gu.seeded_normal((F_, 6, 192), 4).to(dev)
    ids = m.dvae.tokenize(img, one_hot=False).flatten(1, 2)                 # [F,1024]
    res = {'frames': F_, 'config': 'steve_physion_params.py sizes: 1024 tokens, vocab 4096, d=192, 4 heads, 4 layers'}
    res['dvae_tokenize_ms'] = 1e3 * timed(lambda: m.dvae.tokenize(img, one_hot=False), a.reps)
    z = torch.zeros(F_, 4096, 32, 32, device=dev).scatter_(1, ids.view(F_, 1, 32, 32), 1.)
    res['dvae_detokenize_ms'] = 1e3 * timed(lambda: m.dvae.detokenize(z), a.reps)

    def dec():
        lg = m.trans_decoder(slots, ids[:, :-1].contiguous())
        from slotformer_amd import ops
        return ops.cross_entropy(lg.flatten(0, 1), ids.flatten(0, 1).contiguous())
    res['decoder_forward_xent_ms'] = 1e3 * timed(dec, a.reps)
    L, d, V, N = 1024, 192, 4096, 6
    fl = 4 * (2 * L * d * 3 * d + 2 * L * d * d + 2 * L * L * d        # self-attn (causal: half of 4 L^2 d)
              + 2 * L * d * d + 2 * N * d * 2 * d + 4 * L * N * d + 2 * L * d * d   # cross-attn
              + 4 * L * d * 4 * d) + 2 * L * d * V
    res['decoder_gflop_per_frame'] = fl / 1e9
    res['decoder_tflops'] = fl * F_ / (res['decoder_forward_xent_ms'] * 1e-3) / 1e12
    res['decoder_frames_per_s'] = F_ / (res['decoder_forward_xent_ms'] * 1e-3)
    # greedy generation of the whole 32x32 token grid (STEVESlotFormer.decode): K/V-cached vs the reference's algorithm
    # (re-run the prefix every step; timed on the first 96 steps and on steps 928..1023 via the forward at those lengths)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gi, _ = m.trans_decoder.generate_cached(slots, steps=1024)
    torch.cuda.synchronize()
    res['generate_cached_1024_s'] = time.perf_counter() - t0
    res['generate_cached_tokens_per_s'] = F_ * 1024 / res['generate_cached_1024_s']
    prefix_cost = 0.0
    for Lp in (1, 256, 512, 768, 1023):   # forward cost at a few prefix lengths -> trapezoid estimate of sum over 1024 steps
        idp = gi[:, :Lp - 1].contiguous() if Lp > 1 else gi[:, :0].contiguous()
        prefix_cost += timed(lambda: m.trans_decoder(slots, idp), 3) * (1024 / 5)
    res['generate_prefix_rerun_1024_s_estimate'] = prefix_cost
    # CPU oracle on a bounded sample
    c = a.cpu_frames
    torch.set_num_threads(16)
    img_c, slots_c, ids_c = img[:c].cpu(), slots[:c].cpu(), ids[:c].cpu()
    t = time.perf_counter()
    oracle.dvae_tokenize(img_c, sd, 'dvae.', one_hot=False)
    res['cpu_dvae_tokenize_ms_per_frame'] = 1e3 * (time.perf_counter() - t) / c
    t = time.perf_counter()
    lg = oracle.steve_decoder_forward(slots_c, ids_c[:, :-1], sd, 4, 4)
    torch.nn.functional.cross_entropy(lg.flatten(0, 1), ids_c.flatten(0, 1))
    res['cpu_decoder_ms_per_frame'] = 1e3 * (time.perf_counter() - t) / c
    res['cpu_threads'] = 16
    # parity at this size on the same sample
    lg_h = m.trans_decoder(slots[:c], ids[:c, :-1].contiguous()).cpu()
    res['decoder_rel_err_vs_oracle'] = float((lg_h - lg).abs().max() / lg.abs().max())
    print(json.dumps(res))


if __name__ == '__main__':
    main()
