"""GPU parity of the whole-path engines (through the reference-shaped nn.Module API) against
the committed golden vectors (outputs of the reference's own classes) and the oracle."""
import numpy as np
import pytest
import torch

import golden_util as gu
import oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-3  # north-star bar: 1e-3 relative, fp32


def rel_err(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max()).item()


class slot_chain:
    """with slot_chain(0): the slot branch of batched encodes as per-iteration launches (the forms whose bit identity the older tests pin)"""

    def __init__(self, on):
        self.on = int(on)

    def __enter__(self):
        from slotformer_amd import _lib
        self.lib = _lib.lib()
        self.old = self.lib.sf_get_slot_chain()
        self.lib.sf_set_slot_chain(self.on)

    def __exit__(self, *exc):
        self.lib.sf_set_slot_chain(self.old)


def elementwise_close(a, b, rtol=RTOL, floor=1e-4):
    """north star's 1e-3 rel, element by element in the allclose form: |a - b| <= rtol |b| + floor max|b| (slots cross zero: an absolute floor)"""
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return bool(((a - b).abs() <= rtol * b.abs() + floor * b.abs().max()).all())


def build(cfg, golden, seed, dev, vp=False):
    from slotformer_amd.base_slots import build_model as bb
    from slotformer_amd.video_prediction import build_model as bv
    shapes = gu.shapes_from_golden(golden)
    if vp:
        import os, tempfile  # noqa: E401
        scfg = gu.savi_cfg(cfg['resolution'][0], cfg['slot_dict']['num_slots'], slot_size=cfg['slot_dict']['slot_size'])
        scfg['dec_dict'] = {k: v for k, v in cfg['dec_dict'].items() if k != 'dec_ckp_path'}
        savi = bb(gu.ParamsView(scfg))
        path = os.path.join(tempfile.mkdtemp(), 'savi.pth')
        torch.save({'state_dict': savi.state_dict()}, path)
        full = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
        full['dec_dict']['dec_ckp_path'] = path
        m = bv(gu.ParamsView(full))
    else:
        full = dict(cfg)
        if cfg['model'] == 'STEVE' and 'dvae_dict' not in cfg:
            full.update(dvae_dict=dict(down_factor=4, vocab_size=64, dvae_ckp_path=''),
                        dec_dict=dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64),
                        loss_dict=dict(use_img_recon_loss=False))
        m = bb(gu.ParamsView(full))
    own = {k: v for k, v in m.state_dict().items()}
    golden_keys = {k for k, _ in shapes}
    # the encoder-only STEVE fixture lists the hot-path keys; the image side (dvae.*, trans_decoder.*) keeps its init
    hot = {k: v for k, v in own.items() if k in golden_keys or not k.startswith(('dvae.', 'trans_decoder.'))}
    assert [(k, tuple(v.shape)) for k, v in hot.items()] == shapes, 'state-dict keys/shapes differ from the reference'
    sd = gu.seeded_state_dict(shapes, seed, keep=own)
    m.load_state_dict(sd, strict=len(hot) == len(own))
    return m.eval().to(dev), sd


@pytest.mark.parametrize('name,cfg,B,T,seed,noise_seed', [
    ('savi_c1', gu.C1_SAVI, 2, 3, 101, None),
    ('savi_c1_it3', gu.C1_SAVI_IT3, 1, 2, 102, None),
    ('savi_c2', gu.C2_SAVI, 2, 3, 103, 7),
    ('savi_c5', gu.C5_SAVI, 2, 1, 105, None),
])
@torch.no_grad()
def test_savi_golden(dev, name, cfg, B, T, seed, noise_seed):
    g = gu.load_golden(name)
    m, sd = build(cfg, g, seed, dev)
    m.testing = True
    img = gu.seeded_img(B, T, cfg['resolution'][0]).to(dev)
    data = {'img': img}
    if noise_seed is not None:
        N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
        data['noise'] = gu.seeded_normal((B, T, N, D), noise_seed).to(dev)
    out = m(data)
    assert rel_err(out['post_slots'], g['post_slots']) < RTOL
    assert rel_err(out['kernel_dist'], g['kernel_dist']) < RTOL
    # tighter: fp32 reorder noise only
    assert rel_err(out['post_slots'], g['post_slots']) < 5e-5
    # and element by element, as the rollout fixtures are held (test_rollout_golden)
    assert elementwise_close(out['post_slots'], g['post_slots']) and elementwise_close(out['kernel_dist'], g['kernel_dist'])


@pytest.mark.parametrize('name,cfg,B,T,seed,noise_seed', [
    ('savi_c1', gu.C1_SAVI, 2, 3, 101, None),
    ('savi_c2', gu.C2_SAVI, 2, 3, 103, 7),
])
@torch.no_grad()
def test_savi_golden_conv_fp16x2_optin(dev, name, cfg, B, T, seed, noise_seed):
    """The same fixtures under the OPT-IN convolution arithmetic (two fp16 products, weights rounded to fp16; profiles/r03_probes.txt
    section 14): inside the 1e-3 bar with a stated margin (measured 6.1e-5 / 4.7e-5), outside the default arithmetic's 5e-5."""
    from slotformer_amd import _lib
    lib = _lib.lib()
    assert lib.sf_get_conv_fp16x2() == 0, 'the suite runs on the default arithmetic; SF_CONV_FP16X2 must not be set'
    g = gu.load_golden(name)
    m, sd = build(cfg, g, seed, dev)
    m.testing = True
    data = {'img': gu.seeded_img(B, T, cfg['resolution'][0]).to(dev)}
    if noise_seed is not None:
        N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
        data['noise'] = gu.seeded_normal((B, T, N, D), noise_seed).to(dev)
    lib.sf_set_conv_fp16x2(1)
    try:
        out = m(data)
        torch.cuda.synchronize()
    finally:
        lib.sf_set_conv_fp16x2(0)
    e = rel_err(out['post_slots'], g['post_slots'])
    print(name, 'fp16x2 rel err', e)
    assert e < 1.5e-4
    assert rel_err(m(data)['post_slots'], g['post_slots']) < 5e-5


@torch.no_grad()
def test_savi_chunked_golden(dev):
    """Long-video path (savi.py:431-463): golden produced by the reference's own chunking."""
    g = gu.load_golden('savi_c1_chunked')
    m, sd = build(gu.C1_SAVI, g, 106, dev)
    m.testing = True
    m.clip_len = 1
    img = gu.seeded_img(1, 5, 64).to(dev)
    out = m({'img': img})
    assert rel_err(out['post_slots'], g['post_slots']) < 5e-5
    # explicit chunks carrying prev_slots + predictor state reproduce it too
    m._reset_rnn()
    o1 = m._forward(img[:, :2], None)
    o2 = m._forward(img[:, 2:], o1['post_slots'][:, -1].clone())
    both = torch.cat([o1['post_slots'], o2['post_slots']], 1)
    assert rel_err(both, g['post_slots']) < 5e-5


@torch.no_grad()
def test_steve_golden_and_masks(dev):
    g = gu.load_golden('steve_c4')
    m, sd = build(gu.C4_STEVE, g, 104, dev)
    m.testing = True
    out = m({'img': gu.seeded_img(1, 2, 128).to(dev)})
    assert rel_err(out['slots'], g['slots']) < 5e-5
    assert elementwise_close(out['slots'], g['slots'])
    masks = out['masks'].cpu()
    assert masks.shape == g['masks'].shape
    assert (masks - torch.from_numpy(g['masks'])).abs().max() < 1e-5
    # bit-exact argmax masks wherever the reference's own top-1/top-2 margin exceeds fp32 noise
    am = masks.argmax(2).to(torch.uint8)
    safe = torch.from_numpy(g['margin']) > 1e-5
    assert torch.equal(am[safe], torch.from_numpy(g['argmax'])[safe])
    n_unsafe = int((~safe).sum())
    n_diff = int((am != torch.from_numpy(g['argmax'])).sum())
    print(f'argmax: {n_unsafe} sub-margin pixels of {safe.numel()}, {n_diff} differ')
    assert n_diff <= n_unsafe
    # the escape hatch stays small: the fixture has 19 near-ties among its 32768 mask pixels (0.06 %), none of which differs
    # today -- a regression cannot hide there
    assert n_unsafe <= 32 and n_diff <= 2, (n_unsafe, n_diff)


@pytest.mark.parametrize('B', [1, 5, 33, 70])
@torch.no_grad()
def test_rollout_odd_batches(dev, B):
    """The two-launch rollout layers at awkward batch sizes (row-tile counts that are not multiples of 8, more than 64
    videos) for the sliding-window (C2) and the growing-window single-step (C5) rollouters, against the oracle."""
    m, sd = build(gu.C2_ROLL, gu.load_golden('roll_c2'), 202, dev, vp=True)
    slots = gu.seeded_normal((B, 6, 7, 128), 900 + B)
    ref = oracle.rollouter_forward(slots, 4, sd, gu.C2_ROLL['rollout_dict'])
    assert rel_err(m.rollouter(slots.to(dev), 4), ref) < 2e-4
    m5, sd5 = build(gu.C5_ROLL, gu.load_golden('roll_c5'), 205, dev, vp=True)
    s1 = gu.seeded_normal((B, 1, 8, 128), 800 + B)
    ref5 = oracle.single_step_rollouter_forward(s1, 8, sd5, gu.C5_ROLL['rollout_dict'])
    assert rel_err(m5.rollouter(s1.to(dev), 8), ref5) < 2e-4


@torch.no_grad()
def test_precomputed_cnn_features(dev):
    """sf_savi_cnn_f32 + sf_savi_encode_pre_f32: encoding with the CNN features of the first time steps computed ahead
    (the bench's work stealing) is bit-identical to the plain encode."""
    from slotformer_amd import engine
    g = gu.load_golden('savi_c2')
    m, _ = build(gu.C2_SAVI, g, 103, dev)
    img = gu.seeded_img(2, 3, 128).to(dev)
    noise = gu.seeded_normal((2, 3, 7, 128), 7).to(dev)
    ref, kd, _ = engine.savi_encode(m, img, noise=noise)
    for n_pre in (1, 3):
        feat = engine.savi_cnn(m, img, 0, n_pre)
        assert feat.shape == (n_pre, 2, 4096, 64)
        got, kd2, _ = engine.savi_encode(m, img, noise=noise, feat_pre=feat)
        assert torch.equal(got, ref) and torch.equal(kd2, kd)
    assert rel_err(ref, g['post_slots']) < RTOL


@torch.no_grad()
def test_steve_image_side_golden(dev, precision):
    """Row N2 (second half) through the reference-shaped API: dVAE tokens / reconstruction, teacher-forced Transformer
    decoder logits, token cross-entropy and greedy generation vs the reference's own outputs."""
    g = gu.load_golden('steve_tokens')
    cfg = gu.steve_tokens_cfg()
    m, sd = build(cfg, g, 601, dev)
    img = gu.seeded_img(1, 2, 64, seed=602).to(dev)
    flat = img.flatten(0, 1)
    # dVAE
    lg = m.dvae._logits_nhwc(flat.contiguous()).permute(0, 3, 1, 2)
    print('dvae logits', rel_err(lg, g['dvae_logits']))
    assert rel_err(lg, g['dvae_logits']) < 1e-4
    ids = m.dvae.tokenize(flat, one_hot=False).cpu()
    safe = torch.from_numpy(g['dvae_margin']) > 1e-4 * float(np.abs(g['dvae_logits']).max())
    assert torch.equal(ids[safe], torch.from_numpy(g['dvae_ids'])[safe]) and int((~safe).sum()) < 8
    z_hard = torch.zeros(2, 64, 16, 16).scatter_(1, torch.from_numpy(g['dvae_ids']).unsqueeze(1), 1.).to(dev)
    assert rel_err(m.dvae.detokenize(z_hard), g['recon_hard']) < 1e-4
    z_soft = torch.softmax(torch.from_numpy(g['dvae_logits']), 1).to(dev)
    assert rel_err(m.dvae.detokenize(z_soft.unflatten(0, (1, 2))), g['recon_soft'][None]) < 1e-4
    # full forward with the reference's token ids as targets (token_id input of steve.py:283-311)
    m.testing = False
    tok = torch.from_numpy(g['target_token_id']).to(dev).unflatten(0, (1, 2))
    out = m({'img': img, 'token_id': tok})
    print('slots', rel_err(out['slots'], g['slots']), 'logits', rel_err(out['pred_token_id'], g['pred_token_id']))
    assert rel_err(out['slots'], g['slots']) < RTOL
    assert rel_err(out['pred_token_id'], g['pred_token_id']) < RTOL
    loss = m.calc_train_loss({'img': img}, out)['token_recon_loss']
    assert abs(float(loss) - float(g['token_recon_loss'])) < 1e-3 * float(g['token_recon_loss'])
    # tokens computed on the device give the same targets wherever the reference's margin is not a tie
    out2 = m({'img': img})
    same = out2['target_token_id'].cpu() == torch.from_numpy(g['target_token_id'])
    assert int((~same).sum()) <= int((~safe).sum())
    # greedy generation from the reference's slots
    slots = torch.from_numpy(g['slots']).to(dev).flatten(0, 1)
    idx, logits = m.trans_decoder.generate(slots, steps=g['gen_idx'].shape[1])
    assert torch.equal(idx.cpu(), torch.from_numpy(g['gen_idx']))
    assert not logits.is_cuda and rel_err(logits, g['gen_logits']) < RTOL
    # sampled generation (steve_transformer.py:323-326): at a vanishing temperature the draw IS the argmax; at temperature 1 the tokens are
    # draws from softmax(logits): replayed here step by step with the same generator state and torch's own softmax
    i_cold, _ = m.trans_decoder.generate(slots, steps=6, sample=True, temperature=1e-4)
    assert torch.equal(i_cold.cpu(), torch.from_numpy(g['gen_idx'])[:, :6])
    torch.manual_seed(123)
    i_s, l_s = m.trans_decoder.generate(slots, steps=6, sample=True)
    assert i_s.shape == (slots.shape[0], 6) and int(i_s.min()) >= 0 and int(i_s.max()) < m.trans_decoder.vocab_size
    torch.manual_seed(123)
    prefix = torch.zeros((slots.shape[0], 0), dtype=torch.int64, device=dev)
    for _ in range(6):
        lg = m.trans_decoder.forward(slots, prefix)[:, -1]
        prefix = torch.cat((prefix, torch.multinomial(torch.softmax(lg, -1), 1)), 1)
    assert (prefix == i_s).float().mean() > 0.9   # (the HIP softmax and torch's differ in the last bits: a draw at a bin edge may flip)
    # K/V-cached generation = prefix re-run generation (the reference's algorithm), here over 40 tokens
    i1, l1 = m.trans_decoder.generate(slots, steps=40)
    i2, l2 = m.trans_decoder.generate_cached(slots, steps=40)
    assert torch.equal(i1, i2) and rel_err(l2, l1) < 1e-5


@torch.no_grad()
def test_steve_slotformer_golden(dev, tmp_path):
    """STEVESlotFormer through the reference-shaped API: rollout, token logits / losses, and decode (greedy generation of
    the 16x16 token grid, Gumbel-softmax with the reference's captured noise, dVAE detokenisation)."""
    from slotformer_amd.base_slots import build_model as bb
    from slotformer_amd.video_prediction import build_model as bv
    g = gu.load_golden('steve_slotformer')
    steve = bb(gu.ParamsView(gu.steve_tokens_cfg()))
    path = str(tmp_path / 'steve.pth')
    torch.save({'state_dict': steve.state_dict()}, path)
    cfg = gu.steve_slotformer_cfg()
    cfg['dec_dict']['dec_ckp_path'] = path
    m = bv(gu.ParamsView(cfg))
    shapes = gu.shapes_from_golden(g)
    own = dict(m.state_dict())
    assert [(k, tuple(v.shape)) for k, v in own.items()] == shapes
    m.load_state_dict(gu.seeded_state_dict(shapes, 701, keep=own), strict=True)
    m = m.eval().to(dev)
    rd = cfg['rollout_dict']
    T = rd['history_len'] + cfg['loss_dict']['rollout_len']
    slots = gu.seeded_normal((1, T, rd['num_slots'], rd['slot_size']), 702).to(dev)
    img = gu.seeded_img(1, T, 64, seed=703).to(dev)
    out = m({'slots': slots, 'img': img})
    assert rel_err(out['pred_slots'], g['pred_slots']) < RTOL
    assert torch.equal(out['target_token_id'].cpu(), torch.from_numpy(g['target_token_id']))
    assert rel_err(out['pred_token_id'], g['pred_token_id']) < RTOL
    loss = m.calc_train_loss({'slots': slots, 'img': img}, out)
    assert abs(float(loss['slot_recon_loss']) - float(g['slot_recon_loss'])) < 1e-3 * float(g['slot_recon_loss'])
    assert abs(float(loss['img_recon_loss']) - float(g['img_recon_loss'])) < 1e-3 * float(g['img_recon_loss'])
    soft, hard = m.decode(torch.from_numpy(g['pred_slots'][:, 0]).to(dev), gumbel=torch.from_numpy(g['gumbel']))
    print('decode soft', rel_err(soft, g['soft_recon']), 'hard', rel_err(hard, g['hard_recon']))
    assert rel_err(hard, g['hard_recon']) < RTOL and rel_err(soft, g['soft_recon']) < 5 * RTOL
    s2, h2 = m.decode(torch.from_numpy(g['pred_slots'][:, 0]).to(dev))   # noise drawn on the device
    assert torch.equal(h2, hard) and torch.isfinite(s2).all() and s2.shape == hard.shape


@pytest.mark.parametrize('name,cfg,B,pred_len,seed', [
    ('roll_c1', gu.C1_ROLL, 3, 10, 201),
    ('roll_c2', gu.C2_ROLL, 2, 50, 202),
    ('roll_c4', gu.C4_ROLL, 2, 12, 204),
    ('roll_c4_ref', gu.C4_ROLL_REF, 1, 4, 214),
    ('roll_c5', gu.C5_ROLL, 2, 12, 205),
    # BASELINE.json's full horizons: C4 6+40 (8 layers, slot size 192) and C5 1+80 (window growing 8 -> 48 tokens, then
    # sliding; single_step_slotformer.py:49-90), compared over the whole rollout
    ('roll_c4_full', gu.C4_ROLL, 1, 40, 224),
    ('roll_c5_full', gu.C5_ROLL, 1, 80, 225),
])
@torch.no_grad()
def test_rollout_golden(dev, name, cfg, B, pred_len, seed):
    g = gu.load_golden(name)
    m, sd = build(cfg, g, seed, dev, vp=True)
    rd = cfg['rollout_dict']
    slots = gu.seeded_normal((B, rd['history_len'] + pred_len, rd['num_slots'], rd['slot_size']), seed + 1).to(dev)
    m.rollout_len = pred_len
    out = m({'slots': slots})
    e = rel_err(out['pred_slots'], g['pred_slots'])
    # the same comparison element by element: |a - b| / max(|b|, 1e-3 max|b|) (an absolute floor: slots cross zero)
    a_, b_ = out['pred_slots'].detach().cpu().double(), torch.as_tensor(g['pred_slots']).double()
    ee = ((a_ - b_).abs() / b_.abs().clamp_min(1e-3 * b_.abs().max())).max().item()
    print(name, 'rel err (max-norm)', e, ' element-wise with floor 1e-3 max|ref|', ee)
    assert e < RTOL
    assert e < 2e-4
    # north star's 1e-3 rel, element by element in the allclose form: |a - b| <= 1e-3 |b| + 1e-4 max|b|  (`ee` above has its floor at
    # 1e-3 max|b| and so can reach 1000 x the max-norm error by construction: reported, bounded loosely)
    assert bool(((a_ - b_).abs() <= RTOL * b_.abs() + 1e-4 * b_.abs().max()).all()) and ee < 5e-2
    m.loss_decay_factor = 0.9
    losses = m.calc_train_loss({'slots': slots}, out)
    for k, v in zip(g['loss_names'], g['loss_vals']):
        assert abs(float(losses[str(k)]) - float(v)) < 1e-3 * abs(float(v)) + 1e-6


@torch.no_grad()
def test_exact_f32_mode(dev):
    """SF precision 0 (exact f32 MFMA): rollout parity at fp32-reorder level."""
    from slotformer_amd import _lib
    lib = _lib.lib()
    old = lib.sf_get_precision()
    lib.sf_set_precision(0)
    try:
        g = gu.load_golden('roll_c2')
        m, sd = build(gu.C2_ROLL, g, 202, dev, vp=True)
        slots = gu.seeded_normal((2, 56, 7, 128), 203).to(dev)
        m.rollout_len = 50
        assert rel_err(m({'slots': slots})['pred_slots'], g['pred_slots']) < 5e-5
        gs = gu.load_golden('savi_c1')
        s, _ = build(gu.C1_SAVI, gs, 101, dev)
        s.testing = True
        assert rel_err(s({'img': gu.seeded_img(2, 3, 64).to(dev)})['post_slots'], gs['post_slots']) < 2e-5
    finally:
        lib.sf_set_precision(old)


@torch.no_grad()
def test_c2_full_size_vs_oracle(dev):
    """BASELINE config C2 at reduced batch vs the oracle: encode 6 frames then roll 50 steps."""
    scfg, rcfg = gu.C2_SAVI, gu.C2_ROLL
    gs, gr = gu.load_golden('savi_c2'), gu.load_golden('roll_c2')
    savi, ssd = build(scfg, gs, 103, dev)
    sf, fsd = build(rcfg, gr, 202, dev, vp=True)
    savi.testing = True
    B, T, H = 2, 6, 50
    img = gu.seeded_img(B, T, 128, seed=99)
    noise = gu.seeded_normal((B, T, 7, 128), 5)
    post = savi({'img': img.to(dev), 'noise': noise.to(dev)})['post_slots']
    ref_post = oracle.savi_encode(img, ssd, scfg, noise=noise)['post_slots']
    assert rel_err(post, ref_post) < 5e-5
    pred = sf.rollout(post, H)
    ref_pred = oracle.rollouter_forward(ref_post, H, fsd, rcfg['rollout_dict'])
    e = rel_err(pred, ref_pred)
    print('encode+rollout rel err', e)
    assert e < RTOL


@torch.no_grad()
def test_decode_golden_and_postproc(dev):
    """Row N2: StoSAVi.decode vs the reference decoder's outputs; M1: postproc_mask on the decoded masks."""
    from slotformer_amd.video_prediction.vp_utils import postproc_mask
    g = gu.load_golden('decode_c2')
    cfg = gu.savi_cfg(64, 7, kernel_mlp=False, pred='mlp', rnn=False)
    m, sd = build(cfg, g, 401, dev)
    slots = gu.seeded_normal((2, 7, 128), 404).to(dev)
    recon, recons, masks, s2 = m.decode(slots)
    assert recon.shape == (2, 3, 64, 64) and recons.shape == (2, 7, 3, 64, 64) and masks.shape == (2, 7, 1, 64, 64)
    assert s2 is slots
    assert rel_err(recon, g['recon']) < 2e-4
    ref_r, ref_recons, ref_masks = oracle.savi_decode(slots.cpu(), sd, cfg)
    assert rel_err(recons, ref_recons) < 2e-4
    assert (masks.cpu() - ref_masks).abs().max() < 2e-5
    am = masks.argmax(1).squeeze(1).to(torch.uint8).cpu()
    top2 = ref_masks.squeeze(2).topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(am[safe], torch.from_numpy(g['masks_argmax'])[safe])
    pm = postproc_mask(masks.unsqueeze(0)).to(torch.uint8).cpu()[0]
    assert (pm != torch.from_numpy(g['postproc'])[0]).sum() <= (~safe).sum()
    assert torch.allclose(masks.sum(1), torch.ones_like(masks.sum(1)), atol=1e-5)


@torch.no_grad()
def test_decode_golden_128(dev, precision):
    """Row N2 at BASELINE's resolution: StoSAVi.decode at 128 x 128 (four stride-2 transposed convolutions 8 -> 128, 1x1 head, softmax over
    slots) vs the REFERENCE decoder's outputs (decode_c2_128, tools/gen_golden.py decode_128) and the oracle; argmax / postproc_mask
    bit-equal outside a 1e-4 top-2 margin.  Split-bf16 mode runs the fragment kernels (first layer as one GEMM on the broadcast input,
    parity-class deconvolutions with streamed weights, head in the last layer's epilogue), exact-f32 mode the generic path."""
    from slotformer_amd.video_prediction.vp_utils import postproc_mask
    g = gu.load_golden('decode_c2_128')
    cfg = gu.savi_cfg(128, 7, kernel_mlp=False, pred='mlp', rnn=False)
    m, sd = build(cfg, g, 411, dev)
    slots = gu.seeded_normal((2, 7, 128), 414).to(dev)
    recon, recons, masks, _ = m.decode(slots)
    assert recon.shape == (2, 3, 128, 128) and recons.shape == (2, 7, 3, 128, 128) and masks.shape == (2, 7, 1, 128, 128)
    e = rel_err(recon, g['recon'])
    ref_r, ref_recons, ref_masks = oracle.savi_decode(slots.cpu(), sd, cfg)
    e2, e3 = rel_err(recons, ref_recons), (masks.cpu() - ref_masks).abs().max().item()
    print('decode 128', precision, 'recon vs reference', e, 'recons vs oracle', e2, 'masks abs', e3)
    assert e < 2e-4 and e2 < 2e-4 and e3 < 2e-5
    assert (masks.cpu()[:, :, :, ::8, ::8] - torch.from_numpy(g['masks_sample'])).abs().max() < 2e-5
    am = masks.argmax(1).squeeze(1).to(torch.uint8).cpu()
    top2 = ref_masks.squeeze(2).topk(2, dim=1)[0]
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(am[safe], torch.from_numpy(g['masks_argmax'])[safe])
    pm = postproc_mask(masks.unsqueeze(0)).to(torch.uint8).cpu()[0]
    assert (pm != torch.from_numpy(g['postproc'])[0]).sum() <= (~safe).sum()
    # many frames in one call (one chunk) = the same frames two at a time
    many = gu.seeded_normal((9, 7, 128), 415).to(dev)
    many[3:5] = slots
    r9 = m.decode(many)
    assert torch.equal(r9[0][3:5], recon) and torch.equal(r9[2][3:5], masks)


@torch.no_grad()
def test_forward_with_decode_paths(dev):
    """StoSAVi.forward (testing=False) and SlotFormer.rollout(decode=True) produce the reference's dict keys/shapes."""
    gs = gu.load_golden('savi_c1')
    savi, ssd = build(gu.C1_SAVI, gs, 101, dev)
    img = gu.seeded_img(2, 3, 64).to(dev)
    out = savi({'img': img})
    assert set(out) == {'post_slots', 'kernel_dist', 'img', 'post_recon_combined', 'post_recons', 'post_masks'}
    assert out['post_recon_combined'].shape == (2, 3, 3, 64, 64) and out['post_masks'].shape == (2, 3, 6, 1, 64, 64)
    ref = oracle.savi_decode(torch.as_tensor(gs['post_slots']).flatten(0, 1), ssd, gu.C1_SAVI)[0]
    assert rel_err(out['post_recon_combined'].flatten(0, 1), ref) < 5e-4
    losses = savi.calc_train_loss({'img': img}, out)
    assert set(losses) == {'kld_loss', 'post_recon_loss'} and torch.isfinite(losses['post_recon_loss'])
    gr = gu.load_golden('roll_c1')
    sf, _ = build(gu.C1_ROLL, gr, 201, dev, vp=True)
    past = gu.seeded_normal((2, 6, 6, 128), 5).to(dev)
    d = sf.rollout(past, 3, decode=True, with_gt=True)
    assert d['recon_combined'].shape == (2, 9, 3, 64, 64) and d['masks'].shape == (2, 9, 6, 1, 64, 64)
    assert d['slots'].shape == (2, 9, 6, 128)
    sf.use_img_recon_loss = True
    sf.rollout_len = 3
    o = sf({'slots': torch.cat([past, past[:, :3]], 1)})
    assert set(o) == {'recon_combined', 'recons', 'masks', 'pred_slots', 'gt_slots'}


@torch.no_grad()
def test_chunk_loops_and_small_batches(dev):
    """Encoder frame chunks (B > 32), decoder frame chunks (F > 17 at 64x64x7 slots), B = 1, T = 1."""
    g = gu.load_golden('savi_c2')
    cfg = gu.savi_cfg(64, 7, kernel_mlp=False, pred='mlp', rnn=False, kld='none')
    shapes = gu.shapes_from_golden(g)
    m, sd = build(cfg, g, 103, dev)
    m.testing = True
    img = gu.seeded_img(33, 2, 64, seed=77)
    out = m({'img': img.to(dev)})['post_slots']
    ref = oracle.savi_encode(img, sd, cfg)['post_slots']
    assert rel_err(out, ref) < 5e-5
    # batch-size independence: video 32 alone gives the same slots as inside the batch of 33
    one = m({'img': img[32:33].to(dev)})['post_slots']
    assert rel_err(one, ref[32:33]) < 5e-5
    t1 = m({'img': img[:2, :1].to(dev)})['post_slots']
    assert t1.shape == (2, 1, 7, 128) and rel_err(t1, ref[:2, :1]) < 5e-5
    # decoder: 20 slot-frames -> two chunks
    slots = gu.seeded_normal((20, 7, 128), 9)
    recon = m.decode(slots.to(dev))[0]
    assert rel_err(recon, oracle.savi_decode(slots, sd, cfg)[0]) < 2e-4


@torch.no_grad()
def test_rollout_batch_one_and_zero_steps(dev):
    g = gu.load_golden('roll_c2')
    m, sd = build(gu.C2_ROLL, g, 202, dev, vp=True)
    slots = gu.seeded_normal((2, 56, 7, 128), 203)
    pred1 = m.rollout(slots[:1, :6].to(dev), 5)
    assert rel_err(pred1, torch.as_tensor(g['pred_slots'])[:1, :5]) < 2e-4
    assert m.rollout(slots[:, :6].to(dev), 0).shape == (2, 0, 7, 128)
    with pytest.raises(AssertionError):
        m({'slots': slots[:, :20].to(dev)})  # wrong length: rollout_len + history_len != T  (slotformer.py:266-267)


@torch.no_grad()
def test_full_size_properties(dev):
    """BASELINE config C2 at FULL size (B=32, 128x128, 6+50): size-independent properties instead of an oracle run.
    (1) determinism: two runs are bitwise identical; (2) batch independence: a video's slots do not depend on
    what else is in the batch; (3) composition: a 50-step rollout equals 25 steps + 25 steps restarted from the
    last 6 frames; (4) finite outputs."""
    from slotformer_amd.video_prediction.models import SlotRollouter
    gs = gu.load_golden('savi_c2')
    savi, _ = build(gu.C2_SAVI, gs, 103, dev)
    savi.testing = True
    roll = build(gu.C2_ROLL, gu.load_golden('roll_c2'), 202, dev, vp=True)[0].rollouter
    B, T, H = 32, 6, 50
    img = gu.seeded_img(B, T, 128, seed=5).to(dev)
    noise = gu.seeded_normal((B, T, 7, 128), 6).to(dev)
    a = savi({'img': img, 'noise': noise})['post_slots']
    b = savi({'img': img, 'noise': noise})['post_slots']
    assert torch.equal(a, b)
    sub = savi({'img': img[8:12].contiguous(), 'noise': noise[8:12].contiguous()})['post_slots']
    assert torch.equal(sub, a[8:12])
    assert torch.isfinite(a).all()
    p50 = roll(a, H)
    assert torch.equal(p50, roll(a, H))
    p25 = roll(a, 25)
    assert torch.equal(p25, p50[:, :25])
    hist = torch.cat([a, p25], 1)[:, -6:].contiguous()
    assert torch.equal(roll(hist, 25), p50[:, 25:])
    assert torch.isfinite(p50).all() and p50.shape == (B, H, 7, 128)
    # a 2-video batch selects other GEMM tile / split-K configurations (different summation order, same math)
    assert rel_err(roll(a[3:5].contiguous(), 10), p50[3:5, :10].cpu()) < 1e-4


@pytest.mark.parametrize('name,cfg,seed,B,H', [('C4', gu.C4_ROLL, 224, 16, 40), ('C5', gu.C5_ROLL, 225, 64, 80)])
@torch.no_grad()
def test_full_size_properties_c4_c5(dev, name, cfg, seed, B, H):
    """BASELINE configs C4 (Physion, slot size 192, 8 layers, 6+40, B=16) and C5 (PHYRE, single-step rollouter, 1+80,
    B=64) at FULL size and horizon: determinism, batch independence, restart composition (C4: a rollout restarted from its
    own last 6 frames continues bit-identically; C5's growing window has no such restart rule -- prefix consistency
    instead), finite outputs; plus the B=1 golden rows inside the big batch."""
    g = gu.load_golden('roll_c4_full' if name == 'C4' else 'roll_c5_full')
    m, _ = build(cfg, g, seed, dev, vp=True)
    roll = m.rollouter
    rd = cfg['rollout_dict']
    hist, N, C = rd['history_len'], rd['num_slots'], rd['slot_size']
    x = gu.seeded_normal((B, hist, N, C), seed + 50).to(dev)
    # video 0 of the big batch = the burn-in of the B=1 reference fixture: its rows must match the reference's own output
    x[0] = gu.seeded_normal((1, hist + H, N, C), seed + 1)[0, :hist].to(dev)
    full = roll(x, H)
    assert full.shape == (B, H, N, C) and torch.isfinite(full).all()
    assert torch.equal(full, roll(x, H))                                   # determinism
    e = rel_err(full[:1], g['pred_slots'])
    print(name, 'video 0 of the full batch vs the reference fixture over the whole horizon: rel err', e)
    assert e < 2e-4
    half = roll(x, H // 2)
    assert torch.equal(half, full[:, :H // 2])                             # prefix consistency
    sub = roll(x[5:9].contiguous(), 12)                                    # other tile / split configurations: same math
    assert rel_err(sub, full[5:9, :12].cpu()) < 1e-4
    if name == 'C4':
        histb = torch.cat([x, half], 1)[:, -hist:].contiguous()
        assert torch.equal(roll(histb, H - H // 2), full[:, H // 2:])       # restart composition


@torch.no_grad()
def test_rollout_bf16_entry_point_error_is_measured(dev):
    """`sf_rollout_bf16` (single-pass bf16 products = the reference's --fp16 AMP / BASELINE's literal "bf16")
    on the 6+50 path of config C2 against the REFERENCE fixture: the error is an order of magnitude above the split-bf16
    default and outside the 1e-3 parity bar -- the number DESIGN.md quotes for keeping bf16x3 as the product path."""
    import ctypes as C
    from slotformer_amd import engine, _lib
    g = gu.load_golden('roll_c2')
    m, _ = build(gu.C2_ROLL, g, 202, dev, vp=True)
    rd = gu.C2_ROLL['rollout_dict']
    W, N, Cs, H = rd['history_len'], rd['num_slots'], rd['slot_size'], 50
    slots = gu.seeded_normal((2, W + H, N, Cs), 203).to(dev)
    plan = engine.rollouter_plan(m.rollouter)
    lib = _lib.lib()
    ws = engine.workspace(dev, lib.sf_rollout_workspace_bytes(C.byref(plan.struct), 2), ('roll', 'bf16'))
    buf = slots.clone()
    _lib.check(lib.sf_rollout_bf16(C.byref(plan.struct), buf.data_ptr(), 2, W + H, H, ws.data_ptr(), ws.numel(),
                                   torch.cuda.current_stream().cuda_stream))
    e16 = rel_err(buf[:, W:], g['pred_slots'])
    buf = slots.clone()
    engine.rollout(m.rollouter, buf, W, H)
    e3 = rel_err(buf[:, W:], g['pred_slots'])
    # single-pass fp16 (per-call precision 3, a measurement probe: VERDICT r02 -- the CPU emulation of round 2 put it at 5.6e-4
    # after 50 steps, inside the 1e-3 bar with a thin margin; the linear layers run on ONE fp16 MFMA per product)
    buf = slots.clone()
    engine.rollout(m.rollouter, buf, W, H, opts={'precision': 'fp16'})
    ef = rel_err(buf[:, W:], g['pred_slots'])
    print(f'6+50 rollout vs the reference fixture: single-pass bf16 {e16:.2e}, single-pass fp16 {ef:.2e}, split-bf16 (default) {e3:.2e}')
    assert e3 < 2e-4
    assert 1e-3 < e16 < 5e-2, e16      # usable as an option, not as the parity path
    assert e3 < ef < e16 and ef < 5e-3, ef
    assert lib.sf_get_precision() == 1  # the entry points restore the library mode


@pytest.mark.parametrize('cfg,name,seed,res', [(gu.C1_SAVI, 'savi_c1', 101, 64), (gu.C5_SAVI, 'savi_c5', 105, 128)])
@torch.no_grad()
def test_predictor_step_one_launch_matches_the_unfused_chain(dev, cfg, name, seed, res):
    """pred_step.hip (Transformer predictor + LSTM wrapper of a frame in one launch, predictor.py:20-44,76-135) against the
    chain of GEMM / attention / pointwise launches it replaces, on a ragged number of videos (7 x 6 slots: two workgroups)."""
    import ctypes as C
    from slotformer_amd import engine
    m, _ = build(cfg, gu.load_golden(name), seed, dev)
    m.testing = True
    img = gu.seeded_img(7, 4, res).to(dev)
    key = 'slots' if cfg['model'] == 'STEVE' else 'post_slots'
    fused = m({'img': img})[key].clone()
    plan = engine.encoder_plan(m)
    assert bool(plan.struct.pred_packed), 'the packed predictor weights are missing: the one-launch step did not run'
    saved = plan.struct.pred_packed
    plan.struct.pred_packed = C.POINTER(C.c_void_p)()
    try:
        chain = m({'img': img})[key].clone()
    finally:
        plan.struct.pred_packed = saved
    assert not torch.equal(fused, chain)          # two different kernels ran
    assert rel_err(fused, chain.cpu()) < 2e-5     # split-bf16 products in another summation order


@torch.no_grad()
def test_folded_slot_attention_at_width_192_matches_the_kv_path(dev):
    """STEVE on Physion (slot size = encoder width = 192): the one-launch per-pixel chain (pixel_mlp_feat192_kernel) + Slot Attention on
    the normalised features with the key / value projections folded away (savi.py:44-45,66-89), against the k|v GEMM path."""
    from slotformer_amd import engine
    m, _ = build(gu.C4_STEVE, gu.load_golden('steve_c4'), 104, dev)
    m.testing = True
    img = gu.seeded_img(3, 3, 128).to(dev)
    plan = engine.encoder_plan(m)
    assert plan.struct.enc_fc1_p and plan.struct.enc_fc2_p and plan.struct.sa_fold_q_w, 'the folded / packed copies are missing'
    a = m({'img': img})
    saved = plan.struct.enc_fc1_p
    plan.struct.enc_fc1_p = None
    try:
        b = m({'img': img})
    finally:
        plan.struct.enc_fc1_p = saved
    assert not torch.equal(a['slots'], b['slots'])
    assert rel_err(a['slots'], b['slots'].cpu()) < 5e-5
    assert (a['masks'] - b['masks']).abs().max() < 1e-5


@pytest.mark.parametrize('name', ['C2', 'C5', 'C4'])
@torch.no_grad()
def test_forked_encode_is_bit_identical(dev, name):
    with slot_chain(0):   # (the per-iteration forms: the video-stationary slot branch has its own test below)
        _forked_encode_is_bit_identical(dev, name)


def _forked_encode_is_bit_identical(dev, name):
    """engine.savi_encode(side_stream=...) = sf_savi_encode_fork_f32: the image features of all time steps on the calling stream, the slot
    branches one step behind on a second stream (events) -- the same kernels with the same arguments: the same bits as the one-stream
    encode, eager and captured into a hipGraph (two parallel branches), with injected kernel noise (C2), the predictor's LSTM state (C5)
    and STEVE's masks (C4); also with precomputed features of the first steps."""
    from slotformer_amd import engine
    from slotformer_amd.base_slots import build_model
    cfg = {'C2': gu.C2_SAVI, 'C5': gu.C5_SAVI, 'C4': None}[name]
    if name == 'C4':
        cfg = dict(gu.C4_STEVE)
        cfg.update(dvae_dict=dict(down_factor=4, vocab_size=64, dvae_ckp_path=''), dec_dict=dict(dec_type='slate', dec_num_layers=1, dec_num_heads=4, dec_d_model=64),
                   loss_dict=dict(use_img_recon_loss=False))
    torch.manual_seed(31)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    m.testing = True
    B, T = 5, 7   # (7 steps: the ring of four resident Slot-Attention inputs wraps)
    img = gu.seeded_img(B, T, 128, seed=61).to(dev)
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    noise = engine.kernel_noise(m, gu.seeded_normal((B, T, N, D), 62).to(dev), B, T, dev)
    side = torch.cuda.Stream(device=dev)
    if hasattr(m.predictor, 'reset'):
        m.predictor.reset()
    ref = engine.savi_encode(m, img, noise=noise, want_attn=(name == 'C4'), ws_slot='fk0', side_stream=None)
    torch.cuda.synchronize()
    if hasattr(m.predictor, 'reset'):
        m.predictor.reset()
    out = engine.savi_encode(m, img, noise=noise, want_attn=(name == 'C4'), ws_slot='fk1', side_stream=side)
    torch.cuda.synchronize()
    for a, b in zip(out, ref):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    # precomputed features of the first two steps
    feat = engine.savi_cnn(m, img, 0, 2)
    if hasattr(m.predictor, 'reset'):
        m.predictor.reset()
    out2 = engine.savi_encode(m, img, noise=noise, ws_slot='fk1', side_stream=side, feat_pre=feat)
    torch.cuda.synchronize()
    assert torch.equal(out2[0], ref[0])
    # captured: a graph with two parallel branches, replayed twice
    if name == 'C2':
        cap, side2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        with torch.cuda.stream(cap):
            engine.savi_encode(m, img, noise=noise, ws_slot='fk2', side_stream=side2)
            cap.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
                post = engine.savi_encode(m, img, noise=noise, ws_slot='fk2', side_stream=side2)[0]
        torch.cuda.synchronize()
        for _ in range(2):
            post.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(post, ref[0])


@torch.no_grad()
def test_next_step_prologue_at_the_tail_of_the_slot_update(dev):
    with slot_chain(0):   # (the per-iteration forms: the video-stationary slot branch has its own test below)
        _next_step_prologue_at_the_tail_of_the_slot_update(dev)


def _next_step_prologue_at_the_tail_of_the_slot_update(dev):
    """sf_set_encode_fuse_next: the slot prologue of step t + 1 (ResidualMLPPredictor, kernel_dist, sampling, first q; savi.py:393-402) as the tail of
    step t's last matrix-core slot update (1, the default) against the stand-alone launch on every step (0: fp32 thread-per-output products).  The two
    differ by split-bf16 rounding only; both stay inside the fixture tolerance (test_savi_golden runs the default).  B = 1 / 5 / 32 (ragged last
    workgroup: 7 and 35 rows), with and without injected noise, kernel_dist and the post slots compared."""
    from slotformer_amd import engine, _lib
    from slotformer_amd.base_slots import build_model
    lib = _lib.lib()
    cfg = gu.C2_SAVI
    torch.manual_seed(43)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    m.testing = True
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    old = lib.sf_get_encode_fuse_next()
    try:
        for B, T, with_noise in ((5, 4, True), (1, 3, True), (32, 3, False)):
            img = gu.seeded_img(B, T, 128, seed=91 + B).to(dev)
            noise = engine.kernel_noise(m, gu.seeded_normal((B, T, N, D), 92).to(dev), B, T, dev) if with_noise else torch.zeros(B, T, N, D, device=dev)
            outs = {}
            for mode in (0, 1):
                lib.sf_set_encode_fuse_next(mode)
                outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('fn', mode), side_stream=None)
                torch.cuda.synchronize()
            post0, kd0, at0 = outs[0]
            post1, kd1, at1 = outs[1]
            assert not torch.equal(post0[:, 1:], post1[:, 1:]), 'the two settings ran the same kernels: nothing was fused'
            assert torch.equal(post0[:, 0], post1[:, 0]) and torch.equal(kd0[:, 0], kd1[:, 0])   # step 0: the stand-alone prologue either way
            assert rel_err(post1, post0.cpu()) < 2e-5, (B, T, rel_err(post1, post0.cpu()))
            assert rel_err(kd1, kd0.cpu()) < 2e-5, (B, T, rel_err(kd1, kd0.cpu()))
            assert rel_err(at1, at0.cpu()) < 2e-5
    finally:
        lib.sf_set_encode_fuse_next(old)


@pytest.mark.parametrize('name', ['C2', 'C5'])
@torch.no_grad()
def test_batched_encode_is_bit_identical(dev, name):
    with slot_chain(0):   # (the per-iteration forms: the video-stationary slot branch has its own test below)
        _batched_encode_is_bit_identical(dev, name)


def _batched_encode_is_bit_identical(dev, name):
    """The one-stream encode runs the 64 -> 64 convolutions of ALL time steps as one launch per layer (csrc/engine.hip, batched form; the
    weights-stationary kernel on a CU-masked stream) -- against the step-by-step orders (two-branch form on a second stream; precomputed features of
    the first steps): the same bits, with injected kernel noise (C2), the Transformer + LSTM predictor (C5 shapes at T = 4), STEVE-style attention
    maps, B = 1 / 5 / 32, on a plain stream and on a stream with 96 CUs of its own."""
    import ctypes as C
    from slotformer_amd import engine, _lib
    from slotformer_amd.base_slots import build_model
    lib = _lib.lib()
    cfg = {'C2': gu.C2_SAVI, 'C5': gu.C5_SAVI}[name]
    torch.manual_seed(41)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    m.testing = True
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    h = C.c_void_p()
    _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), (C.c_uint * 8)(*([0xffffffff] * 3 + [0] * 5)), 8))
    masked = torch.cuda.ExternalStream(h.value, device=dev)
    side = torch.cuda.Stream(device=dev)
    for B, T in ((5, 4), (1, 3), (32, 2)):
        img = gu.seeded_img(B, T, 128, seed=71 + B).to(dev)
        noise = engine.kernel_noise(m, gu.seeded_normal((B, T, N, D), 72).to(dev), B, T, dev)
        outs = {}
        for mode in ('batched', 'batched_masked', 'two_branches', 'feat_pre'):
            if hasattr(m.predictor, 'reset'):
                m.predictor.reset()
            torch.cuda.synchronize()
            if mode == 'batched_masked':
                with torch.cuda.stream(masked):
                    outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('bt', mode), side_stream=None)
            elif mode == 'feat_pre':
                outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('bt', mode), side_stream=None, feat_pre=engine.savi_cnn(m, img, 0, 2))
            else:
                outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('bt', mode), side_stream=side if mode == 'two_branches' else None)
            torch.cuda.synchronize()
        for mode in ('batched_masked', 'two_branches', 'feat_pre'):
            for a, b in zip(outs['batched'], outs[mode]):
                assert (a is None) == (b is None) and (a is None or torch.equal(a, b)), (name, B, T, mode)


@torch.no_grad()
def test_slot_chain_matches_the_per_iteration_launches(dev):
    """The slot branch of a batched encode as ONE video-stationary launch (csrc/slot_chain.hip: sf_set_slot_chain(1), the default) against the per-iteration
    launches over the batch (0): Slot Attention as split-bf16 products on feature rows kept as bf16 hi | lo there, exact-f32 products on f32 rows
    here -- split-bf16 rounding apart (both inside the fixture tolerance: test_savi_golden runs the default).  B = 1 / 5 / 32, T = 2 .. 6, injected
    noise, carried state (prev_slots: the chunked encode), post slots / kernel distribution / attention maps compared; plain and CU-masked streams."""
    import ctypes as C
    from slotformer_amd import engine, _lib
    from slotformer_amd.base_slots import build_model
    lib = _lib.lib()
    cfg = gu.C2_SAVI
    torch.manual_seed(47)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    m.testing = True
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    h = C.c_void_p()
    _lib.check(lib.sf_stream_create_cu_mask(C.byref(h), (C.c_uint * 8)(*([0xffffffff] * 3 + [0] * 5)), 8))
    masked = torch.cuda.ExternalStream(h.value, device=dev)
    for B, T, with_noise in ((5, 4, True), (1, 6, True), (32, 2, False), (3, 3, True)):
        img = gu.seeded_img(B, T, 128, seed=131 + B).to(dev)
        noise = engine.kernel_noise(m, gu.seeded_normal((B, T, N, D), 132).to(dev), B, T, dev) if with_noise else torch.zeros(B, T, N, D, device=dev)
        prev = gu.seeded_normal((B, N, D), 133).to(dev) if B == 3 else None
        outs = {}
        for mode in (0, 1, 'masked'):
            with slot_chain(0 if mode == 0 else 1):
                if mode == 'masked':
                    with torch.cuda.stream(masked):
                        outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('sc', str(mode)), side_stream=None, prev_slots=prev)
                else:
                    outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('sc', str(mode)), side_stream=None, prev_slots=prev)
                torch.cuda.synchronize()
        post0, kd0, at0 = outs[0]
        post1, kd1, at1 = outs[1]
        assert not torch.equal(post0, post1), 'the two settings ran the same kernels'
        e = (rel_err(post1, post0.cpu()), rel_err(kd1, kd0.cpu()), rel_err(at1, at0.cpu()))
        print(f'slot chain vs per-iteration launches B={B} T={T}: post {e[0]:.2e} kdist {e[1]:.2e} attn {e[2]:.2e}')
        assert e[0] < 2e-5 and e[1] < 2e-5 and e[2] < 2e-5, (B, T, e)
        assert elementwise_close(post1, post0.cpu(), rtol=1e-4, floor=2e-5)
        for a, b in zip(outs[1], outs['masked']):
            assert torch.equal(a, b), (B, T, 'CU-masked stream')


@torch.no_grad()
def test_slot_attention_on_bf16_rows_matches_the_f32_rows(dev):
    """sf_set_slot_attn_planes(1) (default): the Slot-Attention iterations of the encode as split-bf16 16x16x32 MFMA products on feature rows kept as bf16
    hi | lo (sa_attn_planes_kernel) against exact-f32 products on f32 rows (0: sa_attn_tile_kernel) -- the same records, split-bf16 rounding apart.  Every encode
    form (two branches, batched convolutions, precomputed features), B = 1 / 5 / 32, attention maps and kernel distribution compared; the fixtures hold either
    (test_savi_golden runs the default)."""
    from slotformer_amd import engine, _lib
    from slotformer_amd.base_slots import build_model
    lib = _lib.lib()
    cfg = gu.C2_SAVI
    torch.manual_seed(53)
    m = build_model(gu.ParamsView(cfg)).eval().to(dev)
    m.testing = True
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    side = torch.cuda.Stream(device=dev)
    old = lib.sf_get_slot_attn_planes()
    try:
        for B, T, ss in ((5, 4, None), (1, 3, side), (32, 2, None), (3, 3, side)):
            img = gu.seeded_img(B, T, 128, seed=141 + B).to(dev)
            noise = engine.kernel_noise(m, gu.seeded_normal((B, T, N, D), 142).to(dev), B, T, dev)
            outs = {}
            for mode in (0, 1):
                lib.sf_set_slot_attn_planes(mode)
                outs[mode] = engine.savi_encode(m, img, noise=noise, want_attn=True, ws_slot=('sp', mode), side_stream=ss)
                torch.cuda.synchronize()
            e = [rel_err(a, b.cpu()) for a, b in zip(outs[1], outs[0])]
            print(f'Slot Attention on bf16 rows vs f32 rows B={B} T={T}: post {e[0]:.2e} kdist {e[1]:.2e} attn {e[2]:.2e}')
            assert not torch.equal(outs[0][0], outs[1][0]), 'the two settings ran the same kernels'
            assert max(e) < 2e-5, (B, T, e)
            assert elementwise_close(outs[1][0], outs[0][0].cpu(), rtol=1e-4, floor=2e-5)
    finally:
        lib.sf_set_slot_attn_planes(old)
