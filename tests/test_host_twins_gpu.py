"""The C ABI used WITHOUT torch on the data path: weights placed on the GPU with sf_device_alloc / sf_device_upload,
numpy host buffers handed to the *_host twins (include/slotformer_hip.h), results compared with the committed
golden vectors of the reference classes and with the oracle.  (torch appears only inside golden_util / oracle,
to regenerate the seeded weights and the expected values on the CPU.)"""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu
import oracle
from slotformer_amd._lib import check, lib, sf_rollouter, sf_tfm_layer

pytestmark = pytest.mark.gpu


class DeviceArena:
    """Owns device copies of numpy arrays (hipMalloc through the library; freed on close)."""

    def __init__(self):
        self.ptrs = []

    def put(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        p = C.c_void_p()
        check(lib().sf_device_alloc(C.byref(p), a.nbytes))
        check(lib().sf_device_upload(p, a.ctypes.data_as(C.c_void_p), a.nbytes))
        self.ptrs.append(p)
        return p.value

    def close(self):
        for p in self.ptrs:
            check(lib().sf_device_free(p))
        self.ptrs = []


def host(a):
    return a.ctypes.data_as(C.c_void_p)


def test_rollout_host_twin_matches_reference_golden(dev):
    g = gu.load_golden('roll_c1')
    cfg = gu.C1_ROLL
    rd = cfg['rollout_dict']
    N, Cs, d, nl, nh, ffn, W = (rd['num_slots'], rd['slot_size'], rd['d_model'], rd['num_layers'], rd['num_heads'],
                                rd['ffn_dim'], rd['history_len'])
    keep = {k[len('closed::'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('closed::')}
    sd = {k: v.numpy() for k, v in gu.seeded_state_dict(gu.shapes_from_golden(g), 201, keep=keep).items()}
    B, pred_len = 3, 10
    slots = gu.seeded_normal((B, W + pred_len, N, Cs), 202).numpy().copy()
    arena = DeviceArena()
    try:
        layers = (sf_tfm_layer * nl)()
        for i in range(nl):
            p = f'rollouter.transformer_encoder.layers.{i}.'
            for field, key in [('norm1_g', 'norm1.weight'), ('norm1_b', 'norm1.bias'),
                               ('in_proj_w', 'self_attn.in_proj_weight'), ('in_proj_b', 'self_attn.in_proj_bias'),
                               ('out_proj_w', 'self_attn.out_proj.weight'), ('out_proj_b', 'self_attn.out_proj.bias'),
                               ('norm2_g', 'norm2.weight'), ('norm2_b', 'norm2.bias'), ('lin1_w', 'linear1.weight'),
                               ('lin1_b', 'linear1.bias'), ('lin2_w', 'linear2.weight'), ('lin2_b', 'linear2.bias')]:
                setattr(layers[i], field, arena.put(sd[p + key]))
        m = sf_rollouter()
        m.num_slots, m.slot_size, m.d_model, m.num_layers, m.num_heads, m.ffn_dim = N, Cs, d, nl, nh, ffn
        m.norm_first, m.window_len, m.single_step = 1, W, 0
        m.in_proj_w, m.in_proj_b = arena.put(sd['rollouter.in_proj.weight']), arena.put(sd['rollouter.in_proj.bias'])
        m.out_proj_w, m.out_proj_b = arena.put(sd['rollouter.out_proj.weight']), arena.put(sd['rollouter.out_proj.bias'])
        # token PE: enc_t_pe repeated per slot (slotformer.py:103-106); C1 has no slots PE
        m.pe_tok = arena.put(np.repeat(sd['rollouter.enc_t_pe'][0], N, axis=0))
        m.layers = C.cast(layers, C.POINTER(sf_tfm_layer))
        check(lib().sf_rollout_f32_host(C.byref(m), host(slots), B, W + pred_len, pred_len, None, 0, None))
    finally:
        arena.close()
    pred = slots[:, W:]
    e = np.abs(pred - g['pred_slots']).max() / np.abs(g['pred_slots']).max()
    print('host twin rollout rel err vs reference golden', e)
    assert e < 2e-4


def test_slot_attn_iter_host_twin_matches_oracle(dev):
    B, HW, N, D = 2, 4096, 7, 128
    rs = np.random.RandomState(5)
    kv = rs.standard_normal((B, HW, 2 * D)).astype(np.float32)
    q = rs.standard_normal((B, N, D)).astype(np.float32)
    P = lib().sf_slot_attn_num_partials(HW)
    num = np.zeros((B, P, N, D), np.float32)
    den = np.zeros((B, P, N), np.float32)
    attn = np.zeros((B, N, HW), np.float32)
    k_host, v_host = kv.ctypes.data, kv.ctypes.data + 4 * D     # two views of the interleaved (k | v) rows
    eps, scale = 1e-6, D ** -0.5
    check(lib().sf_slot_attn_iter_f32_host(C.c_void_p(k_host), C.c_void_p(v_host), 2 * D, HW * 2 * D, host(q),
                                            host(num), host(den), host(attn), B, HW, N, D, scale, eps, None))
    # savi.py:82-92 restated in float64
    k, v = kv[..., :D].astype(np.float64), kv[..., D:].astype(np.float64)
    logits = scale * np.einsum('bpd,bnd->bpn', k, q.astype(np.float64))
    a = np.exp(logits - logits.max(-1, keepdims=True))
    a = a / a.sum(-1, keepdims=True)
    assert np.abs(attn - a.transpose(0, 2, 1)).max() < 2e-5
    a = a + eps
    upd = np.einsum('bpn,bpd->bnd', a / a.sum(1, keepdims=True), v)
    got = num.astype(np.float64).sum(1) / den.astype(np.float64).sum(1)[..., None]
    assert np.abs(got - upd).max() < 2e-5 * max(1.0, np.abs(upd).max())


def test_kv_producer_matches_oracle(dev, precision):
    from slotformer_amd import ops  # noqa: F401  (loads the library)
    M, C0, C1, D = 4096 + 128, 64, 128, 128
    rs = np.random.RandomState(11)
    t = lambda *s: torch.from_numpy(rs.standard_normal(s).astype(np.float32))  # noqa: E731
    feat = t(M, C0)
    ln0g, ln0b, ln1g, ln1b = 1 + 0.1 * t(C0), 0.1 * t(C0), 1 + 0.1 * t(C1), 0.1 * t(C1)
    w1, b1, w2, b2, wkv = t(C1, C0) / 8, 0.1 * t(C1), t(C1, C1) / 11, 0.1 * t(C1), t(2 * D, C1) / 11
    h = oracle.layer_norm(feat, ln0g, ln0b)
    h = torch.relu(h @ w1.t() + b1) @ w2.t() + b2
    ref = oracle.layer_norm(h, ln1g, ln1b) @ wkv.t()
    d = lambda x: x.to(dev).contiguous()  # noqa: E731
    args = [d(x) for x in (feat, ln0g, ln0b, w1, b1, w2, b2, ln1g, ln1b, wkv)]
    kv = torch.empty(M, 2 * D, device=dev)
    nb = lib().sf_kv_producer_workspace_bytes(M, C1)
    ws = torch.empty(nb // 4, device=dev)
    check(lib().sf_kv_producer_f32(*[a.data_ptr() for a in args], kv.data_ptr(), M, C0, C1, D, 1e-5, ws.data_ptr(), nb,
                                   torch.cuda.current_stream().cuda_stream))
    e = ((kv.cpu() - ref).abs().max() / ref.abs().max()).item()
    print('kv producer rel err', e)
    assert e < 5e-5


def test_kv_producer_host_twin(dev):
    """sf_kv_producer_f32_host: host feature / kv buffers, weights on the device, workspace allocated by the twin."""
    M, C0, C1, D = 4096, 64, 128, 128
    rs = np.random.RandomState(12)
    t = lambda *s: rs.standard_normal(s).astype(np.float32)  # noqa: E731
    feat = t(M, C0)
    ln0g, ln0b, ln1g, ln1b = 1 + 0.1 * t(C0), 0.1 * t(C0), 1 + 0.1 * t(C1), 0.1 * t(C1)
    w1, b1, w2, b2, wkv = t(C1, C0) / 8, 0.1 * t(C1), t(C1, C1) / 11, 0.1 * t(C1), t(2 * D, C1) / 11
    T = torch.from_numpy
    h = oracle.layer_norm(T(feat), T(ln0g), T(ln0b))
    h = torch.relu(h @ T(w1).t() + T(b1)) @ T(w2).t() + T(b2)
    ref = (oracle.layer_norm(h, T(ln1g), T(ln1b)) @ T(wkv).t()).numpy()
    arena = DeviceArena()
    try:
        wd = [arena.put(np.ascontiguousarray(x)) for x in (ln0g, ln0b, w1, b1, w2, b2, ln1g, ln1b, wkv)]
        kv = np.zeros((M, 2 * D), np.float32)
        check(lib().sf_kv_producer_f32_host(host(feat), *wd, host(kv), M, C0, C1, D, 1e-5, None, 0, None))
    finally:
        arena.close()
    e = np.abs(kv - ref).max() / np.abs(ref).max()
    print('kv producer host twin rel err', e)
    assert e < 5e-5


def test_slot_update_host_twin(dev):
    """sf_slot_update_f32_host vs savi.py:95-100 restated with torch's GRUCell / LayerNorm on the CPU."""
    B, N, D, H, P = 3, 7, 128, 256, 16
    rs = np.random.RandomState(13)
    t = lambda *s: rs.standard_normal(s).astype(np.float32)  # noqa: E731
    num, den = t(B, P, N, D), np.abs(t(B, P, N)) + 0.5
    prev = t(B, N, D)
    gru = torch.nn.GRUCell(D, D)
    ln = torch.nn.LayerNorm(D)
    fc1, fc2 = torch.nn.Linear(D, H), torch.nn.Linear(H, D)
    with torch.no_grad():
        upd = torch.from_numpy(num.sum(1) / den.sum(1)[..., None])
        s = gru(upd.reshape(-1, D), torch.from_numpy(prev).reshape(-1, D))
        ref = (s + fc2(torch.relu(fc1(ln(s))))).reshape(B, N, D).numpy()
    tr = lambda w: np.ascontiguousarray(w.detach().numpy().T)  # noqa: E731  ([in, out] layout of the slot-update kernel)
    vec = lambda w: np.ascontiguousarray(w.detach().numpy())  # noqa: E731
    arena = DeviceArena()
    try:
        wd = [arena.put(x) for x in (tr(gru.weight_ih), tr(gru.weight_hh), vec(gru.bias_ih), vec(gru.bias_hh), vec(ln.weight),
                                     vec(ln.bias), tr(fc1.weight), vec(fc1.bias), tr(fc2.weight), vec(fc2.bias))]
        out = np.zeros((B, N, D), np.float32)
        check(lib().sf_slot_update_f32_host(host(num), host(den), P, host(prev), *wd, host(out), B, N, D, H, 1e-5, None))
    finally:
        arena.close()
    e = np.abs(out - ref).max() / np.abs(ref).max()
    print('slot update host twin rel err', e)
    assert e < 2e-5
