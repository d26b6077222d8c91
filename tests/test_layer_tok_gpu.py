"""Token-stationary whole-layer launches (csrc/layer_tok.hip; sf_rollout_opts.layer_tok): every product of a pre-LN nn.TransformerEncoderLayer
(slotformer.py:72-80) as one chain of register-resident MFMA products per 32-token wave.  Kernel level against a float64 PyTorch restatement of the
layer; the rollout with them against the REFERENCE's fixtures; properties (other sequences of the call do not matter, determinism)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import golden_util as gu

pytestmark = pytest.mark.gpu


def _rollouter(dev, seed=3):
    from slotformer_amd.video_prediction.models import SlotRollouter
    torch.manual_seed(seed)
    r = SlotRollouter(**dict(gu.C2_ROLL['rollout_dict'])).eval().to(dev)
    with torch.no_grad():   # biases / LayerNorm parameters away from their (zero / one) initial values: a wrong index must show
        for p in r.parameters():
            if p.dim() == 1:
                p.add_(0.3 * torch.randn_like(p))
    return r


def _layer64(layer, x):
    """x [B, L, 256] -> the layer in float64"""
    d = lambda t: t.detach().double()  # noqa: E731
    xx = x.double()
    B, L, D = xx.shape
    h = F.layer_norm(xx, (D, ), d(layer.norm1.weight), d(layer.norm1.bias))
    q, k, v = F.linear(h, d(layer.self_attn.in_proj_weight), d(layer.self_attn.in_proj_bias)).split(D, dim=-1)
    hd = lambda t: t.reshape(B, L, 8, 32).transpose(1, 2)  # noqa: E731
    a = torch.softmax(hd(q) @ hd(k).transpose(-1, -2) / 32 ** 0.5, dim=-1) @ hd(v)
    x2 = xx + F.linear(a.transpose(1, 2).reshape(B, L, D), d(layer.self_attn.out_proj.weight), d(layer.self_attn.out_proj.bias))
    h2 = F.layer_norm(x2, (D, ), d(layer.norm2.weight), d(layer.norm2.bias))
    return x2 + F.linear(F.relu(F.linear(h2, d(layer.linear1.weight), d(layer.linear1.bias))), d(layer.linear2.weight), d(layer.linear2.bias))


# (videos, tokens per video): three videos per 128-token workgroup with two padding rows (42), 20 padding rows (36), two videos (48, 64), one key block per
# video (8, 16, 32), a last workgroup with one / two videos, a single video; ONE video per workgroup where two would put a wave's keys in four key blocks
# (50) and for windows of 65..96 tokens (70, 96, and 90 = the reference's Physion window of 15 frames x 6 slots: the last wave idle)
@pytest.mark.parametrize('B,L', [(128, 42), (64, 36), (37, 48), (256, 8), (35, 16), (9, 32), (5, 64), (3, 42), (1, 42), (2, 7), (4, 24), (7, 40),
                                 (4, 50), (2, 70), (33, 90), (3, 96), (1, 65)])
@pytest.mark.parametrize('nl', [1, 3])
@torch.no_grad()
def test_layer_tok_block_vs_float64(dev, B, L, nl):
    from slotformer_amd import _lib, engine
    lib = _lib.lib()
    r = _rollouter(dev)
    plan = engine.rollouter_plan(r)
    st = torch.cuda.current_stream().cuda_stream
    x = gu.seeded_normal((B, L, 256), 100 * B + L).to(dev)
    ref = x
    for k in range(nl):
        ref = _layer64(r.transformer_encoder.layers[k], ref)
    y = torch.full((B, L, 256), float('nan'), device=dev)
    _lib.check(lib.sf_layer_tok_block_f32(plan.struct.layers, nl, x.data_ptr(), y.data_ptr(), B, L, st))
    torch.cuda.synchronize()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    print(f'B {B} L {L} layers {nl}: rel err vs float64 {err:.2e}')
    assert err < 1e-5      # (split-bf16 products: 2e-6 per layer measured)
    # the other sequences of a call do not matter (a workgroup holds 128 // L of them; keys of other sequences are masked): replace all but one
    if B >= 2:
        x2 = gu.seeded_normal((B, L, 256), 7).to(dev)
        keep = B // 2
        x2[keep] = x[keep]
        y2 = torch.empty_like(y)
        _lib.check(lib.sf_layer_tok_block_f32(plan.struct.layers, nl, x2.data_ptr(), y2.data_ptr(), B, L, st))
        torch.cuda.synchronize()
        assert torch.equal(y2[keep], y[keep])
    y3 = torch.empty_like(y)
    _lib.check(lib.sf_layer_tok_block_f32(plan.struct.layers, nl, x.data_ptr(), y3.data_ptr(), B, L, st))
    torch.cuda.synchronize()
    assert torch.equal(y3, y)


@torch.no_grad()
def test_layer_tok_argument_errors(dev):
    from slotformer_amd import _lib, engine
    lib = _lib.lib()
    r = _rollouter(dev)
    plan = engine.rollouter_plan(r)
    x = torch.zeros(2, 97, 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.sf_layer_tok_block_f32(plan.struct.layers, 1, x.data_ptr(), x.data_ptr(), 2, 97, st) != 0     # more than 96 tokens per sequence
    assert lib.sf_layer_tok_block_f32(plan.struct.layers, 9, x.data_ptr(), x.data_ptr(), 2, 42, st) != 0     # more than 8 layers per launch
    assert lib.sf_layer_tok_block_f32(plan.struct.layers, 0, x.data_ptr(), x.data_ptr(), 2, 42, st) != 0     # no layer
    assert lib.sf_rollout_tok_ok(C.byref(plan.struct)) == 1
    assert lib.sf_layer_tok_packed_bytes() == 96 * 32768 + 3328 * 4


@pytest.mark.parametrize('name,cfg,B,pred_len,seed', [('roll_c2', gu.C2_ROLL, 2, 50, 202), ('roll_c4_full', gu.C4_ROLL, 1, 40, 224), ('roll_c5_full', gu.C5_ROLL, 1, 80, 225),
                                                      ('roll_c4_ref', gu.C4_ROLL_REF, 1, 4, 214)])
@torch.no_grad()
def test_rollout_with_layer_tok_vs_reference_fixture(dev, name, cfg, B, pred_len, seed):
    """The rollout with the layers before the last as token-stationary launches against the reference's own outputs: C2 (projection ring, sliding window
    of 42 tokens), C4 (slot size 192: no ring, 8 layers -> seven layers in ONE launch, 36 tokens), C5 (single-step rollouter: the window grows 8 -> 48),
    C4 with the reference's own Physion window (slotformer_physion_params.py: 15 burn-in frames x 6 slots = 90 tokens: one video per workgroup)."""
    from test_engine_gpu import build
    from slotformer_amd import engine
    g = gu.load_golden(name)
    m, _ = build(cfg, g, seed, dev, vp=True)
    rd = cfg['rollout_dict']
    hist, N, Cs = rd['history_len'], rd['num_slots'], rd['slot_size']
    T_in = engine.burn_in_of(m.rollouter)
    slots = gu.seeded_normal((B, hist + pred_len, N, Cs), seed + 1).to(dev)
    buf = torch.zeros(B, T_in + pred_len, N, Cs, device=dev)
    buf[:, :T_in] = slots[:, :T_in]
    engine.rollout(m.rollouter, buf, T_in, pred_len, opts={'layer_tok': True})
    ref = torch.as_tensor(g['pred_slots']).to(dev)
    e = ((buf[:, T_in:] - ref).abs().max() / ref.abs().max()).item()
    other = buf.clone()
    other[:, T_in:] = 0
    engine.rollout(m.rollouter, other, T_in, pred_len, opts={'layer_tok': False})
    d = ((buf - other).abs().max() / ref.abs().max()).item()
    print(name, 'token-stationary layers vs the reference fixture', e, ' vs the default forms', d)
    assert e < 5e-5 and 0 < d < 5e-5   # (the reference fixtures: 5e-5 like the encode fixtures; measured 9e-6 .. 3.5e-5)
