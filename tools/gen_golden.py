#!/usr/bin/env python
"""Generate golden vectors by running the REFERENCE's own classes on CPU.

Runs only in the authoring container (needs /root/reference).  The reference
source never travels: this script writes *outputs only* (plus the reference
state-dict key/shape list and closed-form buffers) to tests/golden/*.npz.
Weights and inputs are regenerated from seeds via tests/golden_util.py.

The un-vendored third-party package ``nerv`` (v0.1.0) is replaced by a minimal
in-memory stand-in for the symbols the model files import:
  nerv.training.BaseModel          -> torch.nn.Module
  nerv.models.conv_norm_act        -> Sequential(Conv2d(pad=k//2, bias=True), Identity, ReLU|Identity)
  nerv.models.deconv_norm_act      -> Sequential(ConvTranspose2d(pad=k//2, output_padding=stride-1), Identity, ReLU)
  nerv.models.deconv_out_shape     -> (in-1)*stride - 2*pad + k + out_pad
This convention is NOT pinned by anything in the reference ("parity unpinned").

Also validates oracle/ against the reference on every case and prints max errors.
"""
import os
import sys
import types
import tempfile

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import golden_util as gu  # noqa: E402
import oracle  # noqa: E402


# ---------------------------------------------------------------------------
def install_shims():
    nerv = types.ModuleType('nerv')
    training = types.ModuleType('nerv.training')
    models = types.ModuleType('nerv.models')
    utils = types.ModuleType('nerv.utils')

    class BaseModel(nn.Module):
        pass

    class BaseParams:
        def get(self, k, d=None):
            return getattr(self, k, d)

    def conv_norm_act(cin, cout, kernel_size, stride=1, norm='', act='relu', **kw):
        assert norm == ''
        return nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=kernel_size // 2, bias=True),
            nn.Identity(), nn.ReLU() if act == 'relu' else nn.Identity())

    def deconv_norm_act(cin, cout, kernel_size, stride=1, norm='', act='relu', **kw):
        assert norm == ''
        return nn.Sequential(
            nn.ConvTranspose2d(cin, cout, kernel_size, stride=stride, padding=kernel_size // 2,
                               output_padding=stride - 1, bias=True), nn.Identity(),
            nn.ReLU() if act == 'relu' else nn.Identity())

    def deconv_out_shape(in_size, stride, padding, kernel_size, out_padding):
        return (in_size - 1) * stride - 2 * padding + kernel_size + out_padding

    training.BaseModel = BaseModel
    training.BaseParams = BaseParams
    models.conv_norm_act = conv_norm_act
    models.deconv_norm_act = deconv_norm_act
    models.deconv_out_shape = deconv_out_shape
    for name in ('load_obj', 'dump_obj', 'mkdir_or_exist'):
        setattr(utils, name, lambda *a, **k: None)
    nerv.training, nerv.models, nerv.utils = training, models, utils
    sys.modules.update({'nerv': nerv, 'nerv.training': training, 'nerv.models': models,
                        'nerv.utils': utils})
    # stub parent packages so datasets/ (torchvision, pycocotools, phyre) is never imported
    for pkg, path in (('slotformer', 'slotformer'), ('slotformer.base_slots', 'slotformer/base_slots'),
                      ('slotformer.video_prediction', 'slotformer/video_prediction')):
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, path)]
        sys.modules[pkg] = m


install_shims()
from slotformer.base_slots.models import StoSAVi, STEVE, dVAE  # noqa: E402
from slotformer.base_slots.models import build_model as ref_build_base  # noqa: E402
from slotformer.video_prediction.models import build_model as ref_build_vp  # noqa: E402

TMP = tempfile.mkdtemp(prefix='sf_golden_')


def shapes_of(model):
    return [(k, tuple(v.shape)) for k, v in model.state_dict().items()]


def load_seeded(model, seed):
    ref_sd = model.state_dict()
    sd = gu.seeded_state_dict(shapes_of(model), seed, keep=ref_sd)
    model.load_state_dict(sd, strict=True)
    return {k: v.clone() for k, v in model.state_dict().items()}


def pack_meta(model, sd):
    keys = [k for k, _ in shapes_of(model)]
    shp = ['x'.join(str(d) for d in s) for _, s in shapes_of(model)]
    out = dict(sd_keys=np.array(keys), sd_shapes=np.array(shp))
    for k in keys:
        if k.endswith(gu.CLOSED_FORM_SUFFIXES):
            out['closed::' + k] = sd[k].numpy()
    return out


def save(name, **arrs):
    os.makedirs(gu.GOLDEN_DIR, exist_ok=True)
    path = os.path.join(gu.GOLDEN_DIR, name + '.npz')
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrs.items()})
    print(f'  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KB')


def err(a, b):
    return float((a - b).abs().max()), float(((a - b).abs() / (b.abs() + 1e-6)).median())


class InjectedRandn:
    """Replace torch.randn_like inside the reference with a pre-generated noise stream."""

    def __init__(self, noise):  # noise [B,T,N,D]
        self.noise, self.t = noise, 0

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: self._next(x)
        return self

    def _next(self, x):
        n = self.noise[:, self.t]
        self.t += 1
        assert n.shape == x.shape
        return n.clone()

    def __exit__(self, *a):
        torch.randn_like = self.orig


def build_savi(cfg):
    m = ref_build_base(gu.ParamsView(cfg)).eval()
    m.testing = True
    return m


# ---------------------------------------------------------------------------
@torch.no_grad()
def case_savi(name, cfg, B, T, seed, noise_seed=None, sample_stride=97):
    print(name)
    m = build_savi(cfg)
    sd = load_seeded(m, seed)
    img = gu.seeded_img(B, T, cfg['resolution'][0])
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    noise = gu.seeded_normal((B, T, N, D), noise_seed) if noise_seed is not None else None
    if noise is not None:
        with InjectedRandn(noise):
            out = m({'img': img})
    else:
        out = m({'img': img})
    enc = m._get_encoder_out(img.flatten(0, 1))
    o = oracle.savi_encode(img, sd, cfg, noise=noise)
    print('  oracle post_slots err', err(o['post_slots'], out['post_slots']), 'kernel_dist',
          err(o['kernel_dist'], out['kernel_dist']), 'enc',
          err(o['encoder_out'].flatten(0, 1), enc))
    save(name, post_slots=out['post_slots'], kernel_dist=out['kernel_dist'],
         encoder_out_sample=enc[:, ::sample_stride].contiguous(),
         sample_stride=np.int64(sample_stride), **pack_meta(m, sd))
    return m, sd


@torch.no_grad()
def case_savi_chunked(name, cfg, B, T, seed, max_T):
    """Force the reference's long-video chunking (savi.py:431-463) by simulating OOM."""
    print(name)
    m = build_savi(cfg)
    m.clip_len = 1
    sd = load_seeded(m, seed)
    img = gu.seeded_img(B, T, cfg['resolution'][0])
    orig = m._forward

    def limited(img_, prev_slots=None):
        if img_.shape[1] > max_T:
            raise RuntimeError('simulated out of memory')
        return orig(img_, prev_slots)

    m._forward = limited
    out = m({'img': img})
    m._forward = orig
    full = m._forward(img, None)
    print('  ref chunked vs unchunked', err(out['post_slots'], full['post_slots']), 'clip_len',
          m.clip_len)
    o = oracle.savi_forward_chunked(img, sd, cfg, m.clip_len)
    print('  oracle err', err(o['post_slots'], out['post_slots']))
    save(name, post_slots=out['post_slots'], kernel_dist=out['kernel_dist'],
         clip_len=np.int64(m.clip_len), **pack_meta(m, sd))


@torch.no_grad()
def case_steve(name, cfg, B, T, seed):
    print(name)
    dv = dVAE(vocab_size=64, img_channels=3)
    dpath = os.path.join(TMP, 'dvae.pth')
    torch.save({'state_dict': dv.state_dict()}, dpath)
    full = dict(cfg)
    full['dvae_dict'] = dict(down_factor=4, vocab_size=64, dvae_ckp_path=dpath)
    full['dec_dict'] = dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64)
    full['loss_dict'] = dict(use_img_recon_loss=False)
    m = ref_build_base(gu.ParamsView(full)).eval()
    m.testing = True
    # only the encoder-side keys are part of the hot path
    hot = [(k, tuple(v.shape)) for k, v in m.state_dict().items()
           if not k.startswith(('dvae.', 'trans_decoder.'))]
    ref_sd = m.state_dict()
    sd = gu.seeded_state_dict(hot, seed, keep=ref_sd)
    m.load_state_dict(sd, strict=False)
    img = gu.seeded_img(B, T, cfg['resolution'][0])
    out = m({'img': img})
    o = oracle.steve_encode(img, sd, cfg)
    print('  oracle slots err', err(o['slots'], out['slots']), 'masks', err(o['masks'], out['masks']))
    masks = out['masks']
    top2 = masks.topk(2, dim=2)[0]
    margin = (top2[:, :, 0] - top2[:, :, 1])
    print('  min argmax margin', float(margin.min()))
    meta = dict(sd_keys=np.array([k for k, _ in hot]),
                sd_shapes=np.array(['x'.join(str(d) for d in s) for _, s in hot]))
    for k, _ in hot:
        if k.endswith(gu.CLOSED_FORM_SUFFIXES):
            meta['closed::' + k] = sd[k].numpy()
    save(name, slots=out['slots'], masks=masks.to(torch.float32),
         argmax=masks.argmax(2).to(torch.uint8), margin=margin, **meta)


@torch.no_grad()
def case_steve_tokens(name, B, T, seed):
    """Reference STEVE with its image side: dVAE tokens as targets, teacher-forced Transformer-decoder logits and the
    token cross-entropy (steve.py:282-344); greedy generation (steve_transformer.py:305-333); dVAE detokenisation."""
    print(name)
    cfg = gu.steve_tokens_cfg()
    dv = dVAE(vocab_size=cfg['dvae_dict']['vocab_size'], img_channels=3)
    dpath = os.path.join(TMP, 'dvae_tok.pth')
    torch.save({'state_dict': dv.state_dict()}, dpath)
    full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    full['dvae_dict']['dvae_ckp_path'] = dpath
    m = ref_build_base(gu.ParamsView(full)).eval()
    sd = load_seeded(m, seed)
    img = gu.seeded_img(B, T, cfg['resolution'][0], seed=seed + 1)
    m.testing = False
    out = m({'img': img})
    loss = m.calc_train_loss({'img': img}, out)
    slots = out['slots']
    # dVAE alone
    flat = img.flatten(0, 1)
    logits = m.dvae.encoder(flat)
    top2 = logits.topk(2, dim=1)[0]
    margin = top2[:, 0] - top2[:, 1]
    ids = m.dvae.tokenize(flat, one_hot=False)
    z_hard = m.dvae.tokenize(flat, one_hot=True)
    recon_hard = m.dvae.detokenize(z_hard)
    z_soft = torch.softmax(logits, dim=1)
    recon_soft = m.dvae.detokenize(z_soft)
    # greedy generation of the first tokens
    steps = 6
    gen_idx, gen_logits = m.trans_decoder.generate(slots.flatten(0, 1), steps=steps, sample=False)
    g2 = gen_logits.topk(2, dim=-1)[0]
    # oracle agreement
    o = oracle.steve_forward_tokens(img, slots, sd, cfg)
    print('  oracle logits err', err(o['pred_token_id'], out['pred_token_id']), 'loss', float(o['token_recon_loss']),
          float(loss['token_recon_loss']), 'targets equal', bool(torch.equal(o['target_token_id'], out['target_token_id'])))
    print('  oracle dvae logits', err(oracle.dvae_logits(flat, sd, 'dvae.'), logits), 'detok',
          err(oracle.dvae_detokenize(z_soft, sd, 'dvae.'), recon_soft))
    oi, ol = oracle.steve_decoder_generate(slots.flatten(0, 1), steps, sd, cfg['dec_dict']['dec_num_heads'],
                                           cfg['dec_dict']['dec_num_layers'])
    print('  oracle generate ids equal', bool(torch.equal(oi, gen_idx)), 'logits', err(ol, gen_logits),
          'min token margin', float(margin.min()), 'min gen margin', float((g2[..., 0] - g2[..., 1]).min()))
    meta = pack_meta(m, sd)
    for k in list(meta):
        if k.startswith('closed::') and k.endswith(gu.CLOSED_FORM_NOSTORE):
            del meta[k]
    save(name, slots=slots, pred_token_id=out['pred_token_id'], target_token_id=out['target_token_id'],
         token_recon_loss=np.float64(float(loss['token_recon_loss'])), dvae_logits=logits, dvae_margin=margin,
         dvae_ids=ids, recon_hard=recon_hard, recon_soft=recon_soft, gen_idx=gen_idx, gen_logits=gen_logits,
         gen_margin=(g2[..., 0] - g2[..., 1]), **meta)


def case_steve_train(name, B, T, seed, stride=53, img_loss=False):
    """STEVE's own training step in the reference (steve.py:242-351 in train() mode: slots from the encoder side, frozen-dVAE
    token targets, teacher-forced decoder logits, token cross-entropy, backward), dropout probabilities set to 0 (the only
    random part).  Stored like savi_train: loss, per-parameter gradient norm + strided sample."""
    print(name)
    cfg = gu.steve_tokens_cfg()
    with torch.enable_grad():
        dv = dVAE(vocab_size=cfg['dvae_dict']['vocab_size'], img_channels=3)
        dpath = os.path.join(TMP, 'dvae_train.pth')
        torch.save({'state_dict': dv.state_dict()}, dpath)
        full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
        full['dvae_dict']['dvae_ckp_path'] = dpath
        full['loss_dict'] = dict(use_img_recon_loss=img_loss)
        m = ref_build_base(gu.ParamsView(full)).train()
        m.testing = False
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.
        sd = load_seeded(m, seed)
        img = gu.seeded_img(B, T, cfg['resolution'][0], seed=seed + 1)
        extra, gumbel = {}, None
        if img_loss:   # with dropout at p = 0 the Exp(1) draw of gumbel_softmax is the first RNG use of the forward
            hw = cfg['resolution'][0] // cfg['dvae_dict']['down_factor']
            torch.manual_seed(seed + 2)
            expo = torch.empty(B * T, cfg['dvae_dict']['vocab_size'], hw, hw).exponential_()
            gumbel = -(expo + torch.finfo(torch.float32).tiny).log()
            torch.manual_seed(seed + 2)
        out = m({'img': img})
        terms = m.calc_train_loss({'img': img}, out)
        loss = terms['token_recon_loss'] + (terms['img_recon_loss'] if img_loss else 0.)
        loss.backward()
        grads = {n: p_.grad.detach().clone() for n, p_ in m.named_parameters() if p_.grad is not None}
        osd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
        enc = oracle.steve_encode(img, osd, cfg, training=True)
        o = oracle.steve_forward_tokens(img, enc['slots'], osd, cfg, gumbel)
        (o['token_recon_loss'] + (o['img_recon_loss'] if img_loss else 0.)).backward()
        if img_loss:
            print('  img loss', float(terms['img_recon_loss'].detach()), 'oracle', float(o['img_recon_loss'].detach()))
            extra = dict(gumbel=gumbel.numpy(), img_loss=np.array(float(terms['img_recon_loss'].detach())),
                         recon_img=out['recon_img'].detach().numpy())
        print('  loss', float(loss.detach()), 'oracle', float(o['token_recon_loss'].detach()), 'targets equal',
              bool(torch.equal(o['target_token_id'], out['target_token_id'])))
        worst = sorted(((((osd[n].grad - g).norm() / (g.norm() + 1e-30)).item(), n) for n, g in grads.items()), reverse=True)
        print('  oracle grad rel-L2 err (worst 3)', worst[:3], 'params with grad', len(grads))
    names = sorted(grads)
    meta = pack_meta(m, sd)
    for k in list(meta):
        if k.startswith('closed::') and k.endswith(gu.CLOSED_FORM_NOSTORE):
            del meta[k]
    save(name, loss=np.array(float(terms['token_recon_loss'].detach())), stride=np.int64(stride), slots=out['slots'].detach().numpy(),
         target_token_id=out['target_token_id'].numpy(), grad_names=np.array(names), **extra,
         grad_norms=np.array([float(grads[n].norm()) for n in names]),
         **{'gs.' + n: grads[n].flatten()[::stride].numpy() for n in names}, **meta)


def case_dvae_train(name, B, res, vocab, seed, tau=1., hard=False, stride=7):
    """The reference dVAE's own training step (dVAE.py:102-146 in train() mode): Gumbel-softmax relaxed token map, decoder,
    MSE image loss, backward.  The Exp(1) draw inside steve_utils.gumbel_softmax is the first RNG use of the forward, so
    seeding torch right before the call and repeating the draw captures the noise."""
    print(name)
    with torch.enable_grad():
        m = dVAE(vocab_size=vocab, img_channels=3).train()
        sd = load_seeded(m, seed)
        # GroupNorm affine parameters away from (1, 0) so their gradients are exercised through non-trivial values
        img = gu.seeded_img(B, 1, res, seed=seed + 1)[:, 0]
        torch.manual_seed(seed + 2)
        expo = torch.empty(B, vocab, res // 4, res // 4).exponential_()
        gumbel = -(expo + torch.finfo(torch.float32).tiny).log()
        torch.manual_seed(seed + 2)
        out = m({'img': img, 'gumbel_tau': tau, 'hard': hard})
        loss = m.calc_train_loss({'img': img}, out)['recon_loss']
        loss.backward()
        grads = {n: p_.grad.detach().clone() for n, p_ in m.named_parameters() if p_.grad is not None}
        osd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
        o = oracle.dvae_forward_train(img, gumbel, osd, tau, hard)
        o['recon_loss'].backward()
        print('  loss', float(loss.detach()), 'oracle', float(o['recon_loss'].detach()), 'recon err', err(o['recon'].detach(), out['recon'].detach()))
        worst = sorted(((((osd[n].grad - g).norm() / (g.norm() + 1e-30)).item(), n) for n, g in grads.items()), reverse=True)
        print('  oracle grad rel-L2 err (worst 3)', worst[:3], 'params with grad', len(grads))
    names = sorted(grads)
    save(name, loss=np.array(float(loss.detach())), stride=np.int64(stride), tau=np.array(tau), hard=np.array(int(hard)),
         gumbel=gumbel.numpy(), recon=out['recon'].detach().numpy(), z_logits=out['z_logits'].detach().numpy()[:, ::5],
         grad_names=np.array(names), grad_norms=np.array([float(grads[n].norm()) for n in names]),
         **{'gs.' + n: grads[n].flatten()[::stride].numpy() for n in names}, **pack_meta(m, sd))


@torch.no_grad()
def case_steve_slotformer(name, B, seed):
    """Reference STEVESlotFormer: rollout + token loss (steve_slotformer.py:111-161) and decode (:86-103) with the Gumbel
    noise captured (torch.manual_seed before the call; the first RNG draw inside is the Exp(1) tensor)."""
    print(name)
    scfg = gu.steve_tokens_cfg()
    dv = dVAE(vocab_size=scfg['dvae_dict']['vocab_size'], img_channels=3)
    dpath = os.path.join(TMP, 'dvae_sf.pth')
    torch.save({'state_dict': dv.state_dict()}, dpath)
    sfull = {k: (dict(v) if isinstance(v, dict) else v) for k, v in scfg.items()}
    sfull['dvae_dict']['dvae_ckp_path'] = dpath
    steve = ref_build_base(gu.ParamsView(sfull))
    spath = os.path.join(TMP, 'steve_sf.pth')
    torch.save({'state_dict': steve.state_dict()}, spath)
    cfg = gu.steve_slotformer_cfg()
    full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    full['dvae_dict']['dvae_ckp_path'] = dpath
    full['dec_dict']['dec_ckp_path'] = spath
    m = ref_build_vp(gu.ParamsView(full)).eval()
    sd = load_seeded(m, seed)
    rd = cfg['rollout_dict']
    T = rd['history_len'] + cfg['loss_dict']['rollout_len']
    slots = gu.seeded_normal((B, T, rd['num_slots'], rd['slot_size']), seed + 1)
    img = gu.seeded_img(B, T, cfg['resolution'][0], seed=seed + 2)
    out = m({'slots': slots, 'img': img})
    loss = m.calc_train_loss({'slots': slots, 'img': img}, out)
    # decode one frame's slots; `.cuda()` inside decode is a no-op here (CPU reference run)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        torch.manual_seed(777)
        soft, hard = m.decode(out['pred_slots'][:, 0])
    finally:
        torch.Tensor.cuda = orig_cuda
    torch.manual_seed(777)
    V, h, w = cfg['dvae_dict']['vocab_size'], m.h, m.w
    gumbel = -(torch.empty(B, V, h, w).exponential_() + torch.finfo(torch.float32).tiny).log()
    osoft, ohard = oracle.steve_slotformer_decode(out['pred_slots'][:, 0], sd, cfg, gumbel)
    print('  oracle decode soft', err(osoft, soft), 'hard', err(ohard, hard))
    tgt = oracle.dvae_tokenize(img[:, rd['history_len']:].flatten(0, 1), sd, 'dvae.', one_hot=False).flatten(1, 2)
    ol = oracle.steve_decoder_forward(out['pred_slots'].flatten(0, 1), tgt[:, :-1], sd, cfg['dec_dict']['dec_num_heads'],
                                      cfg['dec_dict']['dec_num_layers'], p='decoder.')
    print('  oracle token logits', err(ol, out['pred_token_id']), 'targets equal', bool(torch.equal(tgt, out['target_token_id'])))
    meta = pack_meta(m, sd)
    for k in list(meta):
        if k.startswith('closed::') and k.endswith(gu.CLOSED_FORM_NOSTORE):
            del meta[k]
    save(name, pred_slots=out['pred_slots'], pred_token_id=out['pred_token_id'], target_token_id=out['target_token_id'],
         slot_recon_loss=np.float64(float(loss['slot_recon_loss'])), img_recon_loss=np.float64(float(loss['img_recon_loss'])),
         gumbel=gumbel, soft_recon=soft, hard_recon=hard, **meta)


def build_slotformer(cfg, savi_seed=11):
    scfg = gu.savi_cfg(cfg['resolution'][0], cfg['slot_dict']['num_slots'],
                       slot_size=cfg['slot_dict']['slot_size'])
    scfg['dec_dict'] = {k: v for k, v in cfg['dec_dict'].items() if k != 'dec_ckp_path'}
    savi = build_savi(scfg)
    load_seeded(savi, savi_seed)
    path = os.path.join(TMP, f'savi_{id(cfg)}.pth')
    torch.save({'state_dict': savi.state_dict()}, path)
    full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
    full['dec_dict']['dec_ckp_path'] = path
    return ref_build_vp(gu.ParamsView(full)).eval()


@torch.no_grad()
def case_rollout(name, cfg, B, pred_len, seed, single_step=False):
    print(name)
    m = build_slotformer(cfg)
    sd = load_seeded(m, seed)
    rd = cfg['rollout_dict']
    hist, N, C = rd['history_len'], rd['num_slots'], rd['slot_size']
    slots = gu.seeded_normal((B, hist + pred_len, N, C), seed + 1)
    m.rollout_len = pred_len
    out = m({'slots': slots})
    o = oracle.slotformer_forward(slots, sd, cfg, pred_len, single_step=single_step)
    print('  oracle pred_slots err', err(o['pred_slots'], out['pred_slots']))
    # eval-mode losses (slotformer.py:284-318)
    m.loss_decay_factor = 0.9
    losses = m.calc_train_loss({'slots': slots}, out)
    ol = oracle.slot_mse_losses(o['pred_slots'], o['gt_slots'], training=False, loss_decay_factor=0.9)
    for k in losses:
        assert abs(float(losses[k]) - float(ol[k])) < 1e-5 * max(1, abs(float(losses[k]))), k
    save(name, pred_slots=out['pred_slots'],
         loss_names=np.array(sorted(losses)), loss_vals=np.array([float(losses[k]) for k in sorted(losses)]),
         **pack_meta(m, sd))
    return m, sd


def case_rollout_grads(name, cfg, B, seed, decay=0.9, img=False, pe_seed=None):
    """Gradients of the reference's training loss (SlotFormer.forward + calc_train_loss in train() mode,
    slotformer.py:263-318, then loss.backward()) w.r.t. every rollouter parameter and the burn-in slots.  Dropout is the
    only random part of that path; its probability is set to 0 on the reference modules so the fixture is reproducible
    (the masked arithmetic is pinned separately against the oracle with host-rebuilt masks)."""
    print(name)
    with torch.enable_grad():
        m = build_slotformer(cfg)
        sd = load_seeded(m, seed)
        if pe_seed is not None:   # learnable tables start at zero (slotformer.py:24-26): give the temporal one values, as after some training
            m.rollouter.enc_t_pe.data.copy_(0.1 * gu.seeded_normal(tuple(m.rollouter.enc_t_pe.shape), pe_seed))
            sd = {k: v.clone() for k, v in m.state_dict().items()}
        m.train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.
        rd = cfg['rollout_dict']
        hist, N, C = rd['history_len'], rd['num_slots'], rd['slot_size']
        S = cfg['loss_dict']['rollout_len']
        slots = gu.seeded_normal((B, hist + S, N, C), seed + 1).requires_grad_(True)
        m.loss_decay_factor = decay
        data = {'slots': slots}
        if img:   # image term on: forward decodes the predicted slots (slotformer.py:272-281), loss adds their MSE (:313-326)
            data['img'] = gu.seeded_img(B, hist + S, cfg['resolution'][0], seed + 2)
        out = m(data)
        terms = m.calc_train_loss(data, out)
        loss = terms['slot_recon_loss'] + (terms['img_recon_loss'] if img else 0.)
        loss.backward()
        names = [n for n, p_ in m.named_parameters() if n.startswith('rollouter.') and p_.requires_grad]
        grads = {n: dict(m.named_parameters())[n].grad.detach().clone() for n in names}
        # the oracle under autograd
        osd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
        oslots = slots.detach().clone().requires_grad_(True)
        o = oracle.slotformer_forward(oslots, osd, cfg, S)
        ol = oracle.slot_mse_losses(o['pred_slots'], o['gt_slots'], training=True, loss_decay_factor=decay)['slot_recon_loss']
        if img:
            rec = oracle.savi_decode(o['pred_slots'].flatten(0, 1), osd, cfg)[0].unflatten(0, (B, S))
            ol = ol + ((rec - data['img'][:, hist:])**2).mean()   # no temporal weighting on the image term (:313-326)
        ol.backward()
        print('  loss', float(loss), 'oracle', float(ol))
        print('  oracle grad err', max(err(osd[n].grad, grads[n]) for n in names), 'd_slots', err(oslots.grad, slots.grad))
    save(name, loss=np.array(float(loss)), d_slots=slots.grad.detach().numpy(), pred_slots=out['pred_slots'].detach().numpy(),
         grad_names=np.array(names), **{'grad.' + n: g.numpy() for n, g in grads.items()}, **pack_meta(m, sd))


def case_savi_train(name, cfg, B, T, seed, noise_seed, kld_w=1e-4, stride=53, no_dropout=False):
    """StoSAVi's own training step (scripts/train.py -> SAVi method: forward in train() mode, calc_train_loss savi.py:527-538,
    loss = post_recon_loss + kld_w * kld_loss, backward): the loss terms and, per parameter, the gradient's L2 norm and a
    strided sample (a full gradient set would be megabytes; the oracle, checked here element by element, carries the full
    comparison in the tests)."""
    print(name)
    with torch.enable_grad():
        m = ref_build_base(gu.ParamsView(cfg)).train()
        m.testing = False
        if no_dropout:   # the Transformer predictor's dropout is the only other random part of the step: switched off
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.
        sd = load_seeded(m, seed)
        img = gu.seeded_img(B, T, cfg['resolution'][0], seed + 1)
        N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
        noise = gu.seeded_normal((B, T, N, D), noise_seed) if noise_seed is not None else None
        if noise is not None:
            with InjectedRandn(noise):
                out = m({'img': img})
        else:
            out = m({'img': img})
        terms = m.calc_train_loss({'img': img}, out)
        loss = terms['post_recon_loss'] + kld_w * terms['kld_loss']
        loss.backward()
        grads = {n: p_.grad.detach().clone() for n, p_ in m.named_parameters() if p_.grad is not None}
        # oracle under autograd
        osd = {k: (v.clone().requires_grad_(True) if k in grads else v) for k, v in sd.items()}
        o = oracle.savi_encode(img, osd, cfg, noise=noise)
        rec = oracle.savi_decode(o['post_slots'].flatten(0, 1), osd, cfg)[0].unflatten(0, (B, T))
        okld = oracle.kernel_kld(o['kernel_dist'], cfg)
        ol = ((rec - img)**2).mean() + kld_w * okld
        ol.backward()
        print('  loss', float(loss.detach()), 'oracle', float(ol.detach()), 'kld', float(terms['kld_loss'].detach()), float(okld.detach()))
        worst = sorted(((((osd[n].grad - g).norm() / (g.norm() + 1e-30)).item(), n) for n, g in grads.items()), reverse=True)
        print('  oracle grad rel-L2 err (worst 3; project_q.0.bias has a structurally zero gradient)', worst[:3], 'params with grad', len(grads))
    names = sorted(grads)
    save(name, loss=np.array(float(loss.detach())), post_recon_loss=np.array(float(terms['post_recon_loss'].detach())),
         kld_loss=np.array(float(terms['kld_loss'].detach())), kld_w=np.array(kld_w), stride=np.int64(stride),
         post_slots=out['post_slots'].detach().numpy(), grad_names=np.array(names),
         grad_norms=np.array([float(grads[n].norm()) for n in names]),
         **{'gs.' + n: grads[n].flatten()[::stride].numpy() for n in names}, **pack_meta(m, sd))


@torch.no_grad()
def case_h2(name, cfg, B, seed, frame_offset=2):
    """Run the reference's own rollout_video_slots() (rollout_clevrer_slots.py:20-65) on CPU."""
    print(name)
    m = build_slotformer(cfg)
    sd = load_seeded(m, seed)
    N, C = cfg['rollout_dict']['num_slots'], cfg['rollout_dict']['slot_size']
    pre = {f'v{i}': gu.seeded_normal((128, N, C), seed + 10 + i).numpy() for i in range(B)}
    # load the reference script as a module with its globals patched for CPU
    vp = os.path.join(REF, 'slotformer/video_prediction')
    sys.path.insert(0, vp)
    src = open(os.path.join(vp, 'rollout_clevrer_slots.py')).read()
    mod = types.ModuleType('ref_rollout_clevrer')
    sys.modules.pop('models', None)
    glb = mod.__dict__
    glb['__name__'] = 'ref_rollout_clevrer'
    exec(compile(src, 'rollout_clevrer_slots.py', 'exec'), glb)
    sys.path.remove(vp)
    glb['params'] = types.SimpleNamespace(input_frames=cfg['rollout_dict']['history_len'],
                                          frame_offset=frame_offset)
    glb['tqdm'] = lambda x, **k: x
    wrapper = nn.Module()
    wrapper.module = m
    wrapper.forward = lambda d: m(d)
    old = (torch.cuda.device_count, torch.cuda.empty_cache, torch.Tensor.cuda)
    torch.cuda.device_count = lambda: B
    torch.cuda.empty_cache = lambda: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        res = glb['rollout_video_slots'](wrapper, pre)
    finally:
        torch.cuda.device_count, torch.cuda.empty_cache, torch.Tensor.cuda = old
    ref = torch.from_numpy(np.stack([res[f'v{i}'] for i in range(B)]))
    ori = torch.from_numpy(np.stack([pre[f'v{i}'] for i in range(B)]))
    o = oracle.rollout_video_slots(ori, sd, cfg, frame_offset)
    print('  oracle H2 err', err(o, ref))
    save(name, slots=ref, frame_offset=np.int64(frame_offset), **pack_meta(m, sd))


@torch.no_grad()
def case_decode(name, cfg, Fr, seed):
    print(name)
    m = build_savi(cfg)
    sd = load_seeded(m, seed)
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    slots = gu.seeded_normal((Fr, N, D), seed + 3)
    recon, recons, masks, _ = m.decode(slots)
    o = oracle.savi_decode(slots, sd, cfg)
    print('  oracle decode err', err(o[0], recon), err(o[2], masks))
    sys.path.insert(0, os.path.join(REF, 'slotformer/video_prediction'))
    pm_ref = None
    try:
        src = open(os.path.join(REF, 'slotformer/video_prediction/vp_utils.py')).read()
        # only postproc_mask + FG_THRE are needed; the module imports lpips/skimage at top
        start = src.index('def postproc_mask')
        end = src.index('def masks_to_boxes_w_empty_mask')
        ns = {'torch': torch, 'FG_THRE': 0.5}
        exec(src[start:end], ns)
        pm_ref = ns['postproc_mask'](masks.unsqueeze(0))
    finally:
        sys.path.pop(0)
    pm = oracle.postproc_mask(masks.unsqueeze(0))
    assert torch.equal(pm, pm_ref)
    save(name, recon=recon, masks_argmax=masks.argmax(1).squeeze(1).to(torch.uint8),
         postproc=pm_ref.to(torch.uint8), masks_sample=masks[:, :, :, ::8, ::8].contiguous(),
         **pack_meta(m, sd))


@torch.no_grad()
def case_phyre(name, scfg, rcfg, B, vid_len, seed):
    """test_phyre_planning.py:159-174 composition: SAVi(frame 0) -> zero pad -> SingleStep rollout."""
    print(name)
    savi = build_savi(scfg)
    ssd = load_seeded(savi, seed)
    sf = build_slotformer(rcfg)
    fsd = load_seeded(sf, seed + 5)
    img = gu.seeded_img(B, 1, scfg['resolution'][0])
    slot0 = savi({'img': img})['post_slots']
    Bn, _, N, C = slot0.shape
    slots = torch.zeros((Bn, vid_len, N, C)).type_as(slot0)
    slots[:, :1] = slot0
    sf.rollout_len = vid_len - 1
    out = sf({'slots': slots})
    o = oracle.phyre_encode_rollout(img, ssd, scfg, fsd, rcfg, vid_len)
    print('  oracle H3 err', err(o['pred_slots'], out['pred_slots']))
    meta = {('savi::' + k): v for k, v in pack_meta(savi, ssd).items()}
    meta.update({('sf::' + k): v for k, v in pack_meta(sf, fsd).items()})
    save(name, slot0=slot0, pred_slots=out['pred_slots'], **meta)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == 'steve_tokens':   # regenerate only this fixture
        case_steve_tokens('steve_tokens', B=1, T=2, seed=601)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'steve_slotformer':
        case_steve_slotformer('steve_slotformer', B=1, seed=701)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'steve_train_img':
        case_steve_train('steve_train_img', B=1, T=2, seed=931, img_loss=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'dvae_train':
        case_dvae_train('dvae_train', B=2, res=32, vocab=64, seed=71)
        case_dvae_train('dvae_train_hard', B=1, res=16, vocab=128, seed=73, tau=0.5, hard=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'steve_train':
        case_steve_train('steve_train', B=1, T=2, seed=921)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'savi_train':
        case_savi_train('savi_train', gu.TRAIN_SAVI, B=1, T=2, seed=901, noise_seed=9)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'savi_train_c1':
        case_savi_train('savi_train_c1', gu.C1_SAVI, B=1, T=3, seed=911, noise_seed=None, no_dropout=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'roll_full':   # BASELINE horizons of C4 (6+40) and C5 (1+80), whole rollout
        case_rollout('roll_c4_full', gu.C4_ROLL, B=1, pred_len=40, seed=224)
        case_rollout('roll_c5_full', gu.C5_ROLL, B=1, pred_len=80, seed=225, single_step=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'decode_128':   # the reference decoder at BASELINE's 128 x 128 (four stride-2 layers, savi.py:262-277)
        case_decode('decode_c2_128', gu.savi_cfg(128, 7, kernel_mlp=False, pred='mlp', rnn=False), Fr=2, seed=411)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'roll_train_pe':
        case_rollout_grads('roll_train_pe', gu.TRAIN_ROLL_PE, B=2, seed=821, pe_seed=823)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'roll_train':
        case_rollout_grads('roll_train', gu.TRAIN_ROLL, B=2, seed=801)
        case_rollout_grads('roll_train_img', gu.TRAIN_ROLL_IMG, B=1, seed=811, img=True)
        return
    case_savi('savi_c1', gu.C1_SAVI, B=2, T=3, seed=101)
    case_savi('savi_c1_it3', gu.C1_SAVI_IT3, B=1, T=2, seed=102)
    case_savi('savi_c2', gu.C2_SAVI, B=2, T=3, seed=103, noise_seed=7)
    case_savi('savi_c5', gu.C5_SAVI, B=2, T=1, seed=105)
    case_savi_chunked('savi_c1_chunked', gu.C1_SAVI, B=1, T=5, seed=106, max_T=2)
    case_steve('steve_c4', gu.C4_STEVE, B=1, T=2, seed=104)
    case_rollout('roll_c1', gu.C1_ROLL, B=3, pred_len=10, seed=201)
    case_rollout('roll_c2', gu.C2_ROLL, B=2, pred_len=50, seed=202)
    case_rollout('roll_c4', gu.C4_ROLL, B=2, pred_len=12, seed=204)
    case_rollout('roll_c4_ref', gu.C4_ROLL_REF, B=1, pred_len=4, seed=214)
    case_rollout('roll_c5', gu.C5_ROLL, B=2, pred_len=12, seed=205, single_step=True)
    case_rollout('roll_c4_full', gu.C4_ROLL, B=1, pred_len=40, seed=224)
    case_rollout('roll_c5_full', gu.C5_ROLL, B=1, pred_len=80, seed=225, single_step=True)
    case_h2('harness_h2', gu.C1_ROLL, B=2, seed=301, frame_offset=2)
    case_decode('decode_c2', gu.savi_cfg(64, 7, kernel_mlp=False, pred='mlp', rnn=False), Fr=2, seed=401)
    case_phyre('harness_h3', gu.C5_SAVI, gu.C5_ROLL, B=2, vid_len=5, seed=501)
    case_steve_tokens('steve_tokens', B=1, T=2, seed=601)
    case_steve_slotformer('steve_slotformer', B=1, seed=701)
    case_rollout_grads('roll_train', gu.TRAIN_ROLL, B=2, seed=801)
    case_rollout_grads('roll_train_img', gu.TRAIN_ROLL_IMG, B=1, seed=811, img=True)
    case_rollout_grads('roll_train_pe', gu.TRAIN_ROLL_PE, B=2, seed=821, pe_seed=823)
    case_savi_train('savi_train', gu.TRAIN_SAVI, B=1, T=2, seed=901, noise_seed=9)
    case_savi_train('savi_train_c1', gu.C1_SAVI, B=1, T=3, seed=911, noise_seed=None, no_dropout=True)
    case_steve_train('steve_train', B=1, T=2, seed=921)


if __name__ == '__main__':
    main()
