"""Torch-tensor wrappers over the C ABI building blocks (device memory + stream plumbing only).

Every function requires contiguous float32 tensors on a HIP device and launches on torch's
current stream.  There is no CPU path.
"""
import torch

from ._lib import lib, check


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError('slotformer_amd ops need contiguous float32 tensors on a HIP device '
                               f'(got {type(t).__name__} {getattr(t, "dtype", None)} '
                               f'{getattr(t, "device", None)}); there is no CPU fallback')


def linear(x, weight, bias=None, ln=None, residual=None, relu=False, ln_eps=1e-5):
    """act(LN?(x) @ weight.T + bias) + residual;  x [..., K], weight [N, K]."""
    _chk(x, weight, bias, residual, *(ln or ()))
    K = x.shape[-1]
    N = weight.shape[0]
    M = x.numel() // K
    out = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
    g, b = (ln if ln is not None else (None, None))
    check(lib().sf_linear_f32(_p(x), K, _p(weight), _p(bias), _p(g), _p(b), ln_eps, _p(residual), N, _p(out), N,
                              M, N, K, int(relu), _stream()))
    return out


def layernorm(x, gamma, beta, eps=1e-5):
    _chk(x, gamma, beta)
    out = torch.empty_like(x)
    D = x.shape[-1]
    check(lib().sf_layernorm_f32(_p(x), _p(gamma), _p(beta), _p(out), x.numel() // D, D, eps, _stream()))
    return out


def pack_conv_weight(w):
    """[Cout,Cin,k,k] -> [Cout,k,k,Cin]."""
    _chk(w)
    out = torch.empty(w.shape[0], w.shape[2], w.shape[3], w.shape[1], device=w.device, dtype=torch.float32)
    check(lib().sf_pack_conv_weight_f32(_p(w), _p(out), w.shape[0], w.shape[1], w.shape[2], _stream()))
    return out


def pack_conv_frag(w_packed):
    """[64,5,5,64] (pack_conv_weight) -> split-bf16 copy in MFMA-fragment order (uint8 buffer) for conv5x5_frag."""
    _chk(w_packed)
    Cout, ks, _, Cin = w_packed.shape
    out = torch.empty(lib().sf_conv_frag_bytes(Cout, Cin, ks), device=w_packed.device, dtype=torch.uint8)
    check(lib().sf_pack_conv_frag_weights(_p(w_packed), out.data_ptr(), Cout, Cin, ks, _stream()))
    return out


def pack_deconv_frag(w_packed):
    """[64,5,5,64] (pack_deconv_weight of a ConvTranspose2d weight) -> consumption-ordered split-bf16 fragments (uint8 buffer) for
    deconv5x5s2_frag / deconv5x5s2_head."""
    _chk(w_packed)
    Cout, ks, _, Cin = w_packed.shape
    out = torch.empty(lib().sf_deconv_frag_bytes(Cout, Cin, ks, 2), device=w_packed.device, dtype=torch.uint8)
    check(lib().sf_pack_deconv_frag_weights(_p(w_packed), out.data_ptr(), Cout, Cin, ks, 2, _stream()))
    return out


def deconv5x5s2_frag(x, w_frag, bias, relu=True):
    """x [R,H,W,64] NHWC, w_frag = pack_deconv_frag(...) -> [R,2H,2W,64]: ConvTranspose2d(64, 64, 5, stride 2, padding 2, output_padding 1)."""
    _chk(x, bias)
    R, H, W, _ = x.shape
    out = torch.empty(R, 2 * H, 2 * W, 64, device=x.device, dtype=torch.float32)
    check(lib().sf_deconv5x5s2_frag_f32(_p(x), w_frag.data_ptr(), _p(bias), _p(out), R, H, W, int(relu), _stream()))
    return out


def deconv5x5s2_head(x, w_frag, bias, head_w, head_b):
    """x [R,H,64,64] NHWC -> dec [R, 2H * 128, 4] = head_w [4,64] . relu(deconv(x) + bias) + head_b: the last decoder layer with the 1x1
    output convolution in its epilogue."""
    _chk(x, bias, head_w, head_b)
    R, H, W, _ = x.shape
    dec = torch.empty(R, 4 * H * W, 4, device=x.device, dtype=torch.float32)
    check(lib().sf_deconv5x5s2_head_frag_f32(_p(x), w_frag.data_ptr(), _p(bias), _p(head_w), _p(head_b), _p(dec), R, H, W, _stream()))
    return dec


def conv5x5_frag(x, w_frag, bias, relu=True, add=None):
    """x [F,H,64,64] NHWC, w_frag = pack_conv_frag(...) -> [F,H,64,64] (4-row tiles, streamed weight fragments)."""
    _chk(x, bias, add)
    F_, H, W, Cin = x.shape
    out = torch.empty(F_, H, W, 64, device=x.device, dtype=torch.float32)
    check(lib().sf_conv5x5_frag_f32(_p(x), w_frag.data_ptr(), _p(bias), _p(add), _p(out), F_, H, W, int(relu), _stream()))
    return out


def conv5x5_ws(x, w_frag, bias, relu=True, add=None, n_workgroups=0):
    """conv5x5_frag with the weights stationary in registers (csrc/conv_ws.hip): the same bits; any H; n_workgroups 0 = one per CU of the stream."""
    _chk(x, bias, add)
    F_, H, W, Cin = x.shape
    out = torch.empty(F_, H, W, 64, device=x.device, dtype=torch.float32)
    check(lib().sf_conv5x5_ws_f32(_p(x), w_frag.data_ptr(), _p(bias), _p(add), _p(out), F_, H, W, int(relu), int(n_workgroups), _stream()))
    return out


def pack_deconv_weight(w):
    """ConvTranspose2d weight [Cin,Cout,k,k] -> [Cout,k,k,Cin]."""
    _chk(w)
    out = torch.empty(w.shape[1], w.shape[2], w.shape[3], w.shape[0], device=w.device, dtype=torch.float32)
    check(lib().sf_pack_deconv_weight_f32(_p(w), _p(out), w.shape[0], w.shape[1], w.shape[2], _stream()))
    return out


def conv_transpose2d_nhwc(x, w_packed, bias, stride, relu=True):
    """x [F,H,W,Cin] NHWC, w_packed [Cout,k,k,Cin] -> [F,H*s,W*s,Cout] (padding k//2, output_padding s-1)."""
    _chk(x, w_packed, bias)
    F_, H, W, Cin = x.shape
    Cout, ks = w_packed.shape[0], w_packed.shape[1]
    out = torch.empty(F_, H * stride, W * stride, Cout, device=x.device, dtype=torch.float32)
    check(lib().sf_conv_transpose2d_nhwc_f32(_p(x), _p(w_packed), _p(bias), _p(out), F_, H, W, Cin, Cout, ks, stride,
                                             int(relu), _stream()))
    return out


def pos_embed_table(grid, dense_w, dense_b):
    """grid [1,H,W,4] -> [H*W, C]."""
    grid = grid.reshape(-1, 4).contiguous()
    _chk(grid, dense_w, dense_b)
    out = torch.empty(grid.shape[0], dense_w.shape[0], device=grid.device, dtype=torch.float32)
    check(lib().sf_pos_embed_table_f32(_p(grid), _p(dense_w), _p(dense_b), _p(out), grid.shape[0],
                                       dense_w.shape[0], _stream()))
    return out


def conv2d_first(img, weight, bias, stride, relu=True, add=None):
    """img [F,Cin,H,W] NCHW -> [F,Ho,Wo,Cout] NHWC."""
    _chk(img, weight, bias, add)
    F_, Cin, H, W = img.shape
    Cout, ks = weight.shape[0], weight.shape[2]
    Ho = (H + 2 * (ks // 2) - ks) // stride + 1
    Wo = (W + 2 * (ks // 2) - ks) // stride + 1
    out = torch.empty(F_, Ho, Wo, Cout, device=img.device, dtype=torch.float32)
    check(lib().sf_conv2d_nchw_in_f32(_p(img), Cin * H * W, _p(weight), _p(bias), _p(add), _p(out), F_, Cin, H, W,
                                      Cout, ks, stride, int(relu), _stream()))
    return out


def conv2d_nhwc(x, w_packed, bias, relu=True, add=None):
    """x [F,H,W,Cin] NHWC, w_packed [Cout,k,k,Cin] -> [F,H,W,Cout]."""
    _chk(x, w_packed, bias, add)
    F_, H, W, Cin = x.shape
    Cout, ks = w_packed.shape[0], w_packed.shape[1]
    out = torch.empty(F_, H, W, Cout, device=x.device, dtype=torch.float32)
    check(lib().sf_conv2d_nhwc_f32(_p(x), _p(w_packed), _p(bias), _p(add), _p(out), F_, H, W, Cin, Cout, ks,
                                   int(relu), _stream()))
    return out


def slot_attn_iter(k, v, q, eps=1e-6, want_attn=False):
    """k, v [B,HW,D]; q [B,N,D] -> (part_num [B,P,N,D], part_den [B,P,N], attn [B,N,HW] | None)."""
    _chk(k, v, q)
    B, HW, D = k.shape
    N = q.shape[1]
    P = lib().sf_slot_attn_num_partials(HW)
    pn = torch.empty(B, P, N, D, device=k.device, dtype=torch.float32)
    pd = torch.empty(B, P, N, device=k.device, dtype=torch.float32)
    attn = torch.empty(B, N, HW, device=k.device, dtype=torch.float32) if want_attn else None
    check(lib().sf_slot_attn_iter_f32(_p(k), _p(v), D, HW * D, _p(q), _p(pn), _p(pd), _p(attn), B, HW, N, D,
                                      float(D)**-0.5, eps, _stream()))
    return pn, pd, attn


def slot_attn_iter_bwd(k, v, q, pn, pd, d_updates, dk=None, dv=None, eps=1e-6):
    """Backward of slot_attn_iter: gradient d_updates [B,N,D] of updates = sum(pn) / sum(pd) -> (dq, dk, dv).  Passing
    dk / dv accumulates into them (the iterations of a frame share k and v)."""
    _chk(k, v, q, pn, pd, d_updates)
    B, HW, D = k.shape
    N, P = q.shape[1], pn.shape[1]
    acc = dk is not None
    if not acc:
        dk, dv = torch.empty_like(k), torch.empty_like(v)
    dq = torch.empty_like(q)
    nb = lib().sf_slot_attn_iter_bwd_workspace_bytes(B, HW, N, D)
    ws = torch.empty(nb, dtype=torch.uint8, device=k.device)
    check(lib().sf_slot_attn_iter_bwd_f32(_p(k), _p(v), D, HW * D, _p(q), _p(pn), _p(pd), P, _p(d_updates), _p(dk), _p(dv),
                                          int(acc), _p(dq), B, HW, N, D, float(D)**-0.5, eps, ws.data_ptr(), nb, _stream()))
    return dq, dk, dv


def slot_update(pn, pd, slots_prev, gru, ln_g, ln_b, w1, b1, w2, b2, ln_eps=1e-5):
    """gru = (w_ih, w_hh, b_ih, b_hh); all weights in torch layout (transposed here for the kernel)."""
    _chk(pn, pd, slots_prev, *gru, ln_g, ln_b, w1, b1, w2, b2)
    B, P, N, D = pn.shape
    out = torch.empty_like(slots_prev)
    w_ih_t, w_hh_t, w1_t, w2_t = (w.t().contiguous() for w in (gru[0], gru[1], w1, w2))
    check(lib().sf_slot_update_f32(_p(pn), _p(pd), P, _p(slots_prev), _p(w_ih_t), _p(w_hh_t), _p(gru[2]),
                                   _p(gru[3]), _p(ln_g), _p(ln_b), _p(w1_t), _p(b1), _p(w2_t), _p(b2), _p(out),
                                   B, N, D, w1.shape[0], ln_eps, _stream()))
    return out


def pack_linear(w):
    """torch-layout weight [N, K] -> fragment-ordered split-bf16 copy (sf_pack_linear_weights)."""
    _chk(w)
    n, k = w.shape
    buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
    check(lib().sf_pack_linear_weights(_p(w.contiguous()), buf.data_ptr(), n, k, _stream()))
    return buf


def slot_update_packed(pn, pd, slots_prev, gru, ln_g, ln_b, w1, b1, w2, b2, q=None, ln_eps=1e-5):
    """slot_update on the matrix cores (slot size 128, MLP 256); q = (q_ln_g, q_ln_b, q_w) also returns project_q(out)."""
    _chk(pn, pd, slots_prev, *gru, ln_g, ln_b, w1, b1, w2, b2)
    B, P, N, D = pn.shape
    out = torch.empty_like(slots_prev)
    packed = [pack_linear(w) for w in (gru[0], gru[1], w1, w2)]
    q_out = torch.empty_like(slots_prev) if q is not None else None
    qp = pack_linear(q[2]) if q is not None else None
    check(lib().sf_slot_update_packed_f32(_p(pn), _p(pd), P, _p(slots_prev), packed[0].data_ptr(), packed[1].data_ptr(), _p(gru[2]),
                                          _p(gru[3]), _p(ln_g), _p(ln_b), packed[2].data_ptr(), _p(b1), packed[3].data_ptr(), _p(b2),
                                          _p(out), _p(q[0]) if q is not None else None, _p(q[1]) if q is not None else None,
                                          qp.data_ptr() if qp is not None else None, _p(q_out), B, N, D, w1.shape[0], ln_eps, _stream()))
    return (out, q_out) if q is not None else out


def mha(qkv, B, L, d_model, num_heads, Lq=None):
    """qkv [B*L, 3d] -> [B*Lq, d]."""
    _chk(qkv)
    Lq = L if Lq is None else Lq
    out = torch.empty(B * Lq, d_model, device=qkv.device, dtype=torch.float32)
    check(lib().sf_mha_f32(_p(qkv), _p(out), B, L, Lq, d_model, num_heads, _stream()))
    return out


def qkv_attention(x, in_proj_w, in_proj_b, B, L, num_heads, ln=None, Lq=None, ln_eps=1e-5):
    """Fused LN? -> q|k|v projection -> attention;  x [B*L, d] -> [B*Lq, d] (before out_proj)."""
    _chk(x, in_proj_w, in_proj_b, *(ln or ()))
    d = x.shape[-1]
    Lq = L if Lq is None else Lq
    out = torch.empty(B * Lq, d, device=x.device, dtype=torch.float32)
    g, b = ln if ln is not None else (None, None)
    check(lib().sf_qkv_attention_f32(_p(x), _p(g), _p(b), ln_eps, _p(in_proj_w), _p(in_proj_b), _p(out), B, L, Lq, d,
                                     num_heads, _stream()))
    return out


def lstm_pointwise(gates, c_prev):
    _chk(gates, c_prev)
    R, H4 = gates.shape
    h = torch.empty(R, H4 // 4, device=gates.device, dtype=torch.float32)
    c = torch.empty_like(h)
    check(lib().sf_lstm_pointwise_f32(_p(gates), _p(c_prev), _p(h), _p(c), R, H4 // 4, _stream()))
    return h, c


def sample_dist(dist, noise=None):
    _chk(dist, noise)
    D = dist.shape[-1] // 2
    out = torch.empty(*dist.shape[:-1], D, device=dist.device, dtype=torch.float32)
    check(lib().sf_sample_dist_f32(_p(dist), _p(noise), _p(out), dist.numel() // (2 * D), D, _stream()))
    return out


def bilinear_resize(x, size):
    """x [..., Hi, Wi] -> [..., Ho, Wo]  (F.interpolate bilinear, align_corners=False)."""
    _chk(x)
    Hi, Wi = x.shape[-2:]
    out = torch.empty(*x.shape[:-2], size[0], size[1], device=x.device, dtype=torch.float32)
    check(lib().sf_bilinear_resize_f32(_p(x), _p(out), x.numel() // (Hi * Wi), Hi, Wi, size[0], size[1],
                                       _stream()))
    return out


def groupnorm1_nhwc(x, gamma, beta, eps=1e-5, relu=True, pixel_shuffle=1):
    """F.group_norm(x, 1, gamma, beta) (+ReLU) on NHWC x [F,H,W,C]; pixel_shuffle=2 also applies nn.PixelShuffle(2)
    (-> [F,2H,2W,C/4]).  steve_utils.py:124-126, dVAE.py:44,49."""
    _chk(x, gamma, beta)
    F_, H, W, C_ = x.shape
    r = pixel_shuffle
    out = torch.empty(F_, H * r, W * r, C_ // (r * r), device=x.device, dtype=torch.float32)
    nb = lib().sf_groupnorm1_workspace_bytes(F_)
    ws = torch.empty(max(nb, 8), dtype=torch.uint8, device=x.device)
    check(lib().sf_groupnorm1_nhwc_f32(_p(x), _p(gamma), _p(beta), _p(out), F_, H, W, C_, eps, int(relu), r, ws.data_ptr(), nb,
                                       _stream()))
    return out


def slate_attention(q, k, v, num_heads, causal, q_off=0, k_off=0, v_off=0, d_model=None):
    """softmax(q k^T hd^-0.5 [causal]) v per head (steve_transformer.py:12-55).  q [B,Lq,ldq], k/v [B,Lk,ld*]: the head
    block starts at column *_off of each row (so q|k|v may live side by side in one projection buffer)."""
    _chk(q, k, v)
    B, Lq, ldq = q.shape
    Lk, ldk, ldv = k.shape[1], k.shape[2], v.shape[2]
    d = d_model if d_model is not None else ldq
    out = torch.empty(B, Lq, d, device=q.device, dtype=torch.float32)
    check(lib().sf_slate_attention_f32(q.data_ptr() + 4 * q_off, k.data_ptr() + 4 * k_off, v.data_ptr() + 4 * v_off, _p(out),
                                       ldq, ldk, ldv, d, B, Lq, Lk, num_heads, d // num_heads, int(causal), _stream()))
    return out


def slate_attention_bwd(q, k, v, out, d_out, num_heads, causal):
    """Adjoint of slate_attention for contiguous q [B,Lq,d], k, v [B,Lk,d]: (dq, dk, dv)."""
    _chk(q, k, v, out, d_out)
    B, Lq, d = q.shape
    Lk = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nb = lib().sf_slate_attention_bwd_workspace_bytes(B, Lq, num_heads)
    ws = torch.empty(nb, dtype=torch.uint8, device=q.device)
    check(lib().sf_slate_attention_bwd_f32(_p(q), _p(k), _p(v), _p(out), _p(d_out), _p(dq), _p(dk), _p(dv), d, d, d, d, Lq * d, Lk * d,
                                           Lk * d, Lq * d, B, Lq, Lk, num_heads, d // num_heads, int(causal), ws.data_ptr(), nb, _stream()))
    return dq, dk, dv


def embed_tokens(idx, tok_emb, pos):
    """tok_emb[idx] + pos[:L];  idx int64 [B,L] on device."""
    _chk(tok_emb, pos)
    if idx.dtype != torch.int64 or not idx.is_cuda or not idx.is_contiguous():
        raise TypeError('embed_tokens: idx must be a contiguous int64 device tensor')
    B, L = idx.shape
    d = tok_emb.shape[1]
    out = torch.empty(B, L, d, device=tok_emb.device, dtype=torch.float32)
    check(lib().sf_embed_tokens_f32(idx.data_ptr(), _p(tok_emb), _p(pos), _p(out), B, L, d, _stream()))
    return out


def argmax_rows(x):
    """First index of the maximum of each row of x [..., V] -> int64 [...]."""
    _chk(x)
    V = x.shape[-1]
    R = x.numel() // V
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.int64)
    check(lib().sf_argmax_rows_f32(_p(x), V, out.data_ptr(), R, V, _stream()))
    return out


def cross_entropy(logits, target):
    """F.cross_entropy(logits [R,V], target int64 [R]) with mean reduction -> 0-dim tensor."""
    _chk(logits)
    if target.dtype != torch.int64 or not target.is_cuda or not target.is_contiguous():
        raise TypeError('cross_entropy: target must be a contiguous int64 device tensor')
    R, V = logits.shape
    rows = torch.empty(R, device=logits.device, dtype=torch.float32)
    mean = torch.empty(1, device=logits.device, dtype=torch.float32)
    check(lib().sf_cross_entropy_f32(_p(logits), target.data_ptr(), _p(rows), _p(mean), R, V, _stream()))
    return mean[0]


def softmax_rows(x, add=None, scale=1.0):
    """softmax((x + add) * scale) over the last dim."""
    _chk(x, add)
    V = x.shape[-1]
    out = torch.empty_like(x)
    check(lib().sf_softmax_rows_f32(_p(x), _p(add), float(scale), _p(out), x.numel() // V, V, _stream()))
    return out


def log_softmax_rows(x):
    """F.log_softmax over the last dim."""
    _chk(x)
    V = x.shape[-1]
    out = torch.empty_like(x)
    check(lib().sf_log_softmax_rows_f32(_p(x), _p(out), x.numel() // V, V, _stream()))
    return out


def gumbel_softmax_rows(x, seed, scale=1.0):
    """softmax((x + g) * scale) over the last dim with Gumbel(0, 1) noise g generated inside the kernel from `seed`
    (train.gumbel_noise rebuilds it on the host)."""
    _chk(x)
    V = x.shape[-1]
    out = torch.empty_like(x)
    check(lib().sf_gumbel_softmax_rows_f32(_p(x), int(seed) & 0xffffffffffffffff, float(scale), _p(out), x.numel() // V, V, _stream()))
    return out


def slate_attention_cached(q, kv_cache, Lk, num_heads, d_model, k_off, v_off):
    """One-query-per-sequence attention over the first Lk rows of a K/V cache.  q [B,1,ldq] (head block at column 0),
    kv_cache [B,Lmax,ld] holding k at column k_off and v at column v_off of each row."""
    _chk(q, kv_cache)
    B, Lq, ldq = q.shape
    Lmax, ld = kv_cache.shape[1], kv_cache.shape[2]
    out = torch.empty(B, Lq, d_model, device=q.device, dtype=torch.float32)
    check(lib().sf_slate_attention_strided_f32(q.data_ptr(), kv_cache.data_ptr() + 4 * k_off, kv_cache.data_ptr() + 4 * v_off,
                                               _p(out), ldq, ld, ld, d_model, Lq * ldq, Lmax * ld, Lmax * ld, Lq * d_model, B, Lq,
                                               Lk, num_heads, d_model // num_heads, 0, _stream()))
    return out
