"""Training of the rollout Transformer on the HIP library.

`SlotRollouter.forward` lands here when autograd is recording: the whole autoregressive rollout is ONE autograd node.
Its forward is `sf_rollout_train_fwd_f32` (activations of all steps kept in a workspace owned by the node), its backward
`sf_rollout_train_bwd_f32`, which writes every parameter gradient into one flat fp32 bucket -- the unit the data-parallel
all-reduce works on (parallel.allreduce_flat_bucket's layout, without the pack / unpack copies).  This replaces
torch autograd over the reference's per-step nn.TransformerEncoder calls (slotformer.py:110-124) and nerv's DDP wrap.

Dropout follows nn.TransformerEncoderLayer: active iff the module is in train() mode, p read from the layer.

Data-parallel training: SlotFormer's parameters all belong to ONE node, so `rollouter.ddp_flat_bucket = True` lets that node
all-reduce its bucket inside backward().  StoSAVi's step is a chain of nodes (image encoder, one Slot-Attention node per
frame, predictor layers, decoder) whose gradients autograd accumulates into `.grad`; there the step ends with one
`parallel.allreduce_flat_bucket(model.parameters())` -- still a single RCCL call (4.8 MB).
"""
import ctypes as C

import torch
import torch.nn.functional as tnf

from ._lib import sf_savi_features, sf_savi_features_grads, sf_savi_decoder_grads
from ._lib import lib, check, sf_rollouter_grads, sf_tfm_layer_grads, sf_slot_attention, sf_slot_attention_grads, _SA_LEAVES
from . import engine, parallel

_LAYER_LEAVES = ('norm1.weight', 'norm1.bias', 'self_attn.in_proj_weight', 'self_attn.in_proj_bias',
                 'self_attn.out_proj.weight', 'self_attn.out_proj.bias', 'norm2.weight', 'norm2.bias', 'linear1.weight',
                 'linear1.bias', 'linear2.weight', 'linear2.bias')
_LAYER_FIELDS = ('norm1_g', 'norm1_b', 'in_proj_w', 'in_proj_b', 'out_proj_w', 'out_proj_b', 'norm2_g', 'norm2_b', 'lin1_w',
                 'lin1_b', 'lin2_w', 'lin2_b')


def _leaf(module, dotted):
    for part in dotted.split('.'):
        module = getattr(module, part)
    return module


def rollouter_parameters(r):
    """The trainable leaves of a SlotRollouter in bucket order: in_proj, out_proj, then every layer's leaves."""
    ps = [r.in_proj.weight, r.in_proj.bias, r.out_proj.weight, r.out_proj.bias]
    for layer in r.transformer_encoder.layers:
        ps += [_leaf(layer, n) for n in _LAYER_LEAVES]
    return ps


def learnable_tables(r):
    """The position tables of a rollouter that are trained (t_pe / slots_pe = 'learnable', slotformer.py:19-29 build_pos_enc)."""
    return [p for p in (getattr(r, 'enc_t_pe', None), getattr(r, 'enc_slots_pe', None)) if isinstance(p, torch.nn.Parameter) and p.requires_grad]


def _grad_descriptor(flat, params, num_layers, pe_off=None):
    """sf_rollouter_grads whose pointers are consecutive slices of `flat` (same order as rollouter_parameters); pe_off: where the
    gradient of the folded position table goes (floats into `flat`), None = tables fixed."""
    ptrs, off = [], 0
    for p in params:
        ptrs.append(flat.data_ptr() + 4 * off)
        off += p.numel()
    g = sf_rollouter_grads()
    g.pe_tok = None if pe_off is None else flat.data_ptr() + 4 * pe_off
    g.in_proj_w, g.in_proj_b, g.out_proj_w, g.out_proj_b = ptrs[:4]
    arr = (sf_tfm_layer_grads * num_layers)()
    for i in range(num_layers):
        for j, f in enumerate(_LAYER_FIELDS):
            setattr(arr[i], f, ptrs[4 + i * len(_LAYER_FIELDS) + j])
    g.layers = C.cast(arr, C.POINTER(sf_tfm_layer_grads))
    return g, arr


def _split(flat, params):
    out, off = [], 0
    for p in params:
        out.append(flat[off:off + p.numel()].view_as(p))
        off += p.numel()
    return out


class _Rollout(torch.autograd.Function):
    """pred = rollouter(x, pred_len) as a single autograd node."""

    @staticmethod
    def forward(ctx, r, x, pred_len, p_drop, seed, *params):
        x = x.detach().float().contiguous()
        if not x.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
        plan = engine.rollouter_plan(r, packed=False)
        B = x.shape[0]
        nbytes = lib().sf_rollout_train_workspace_bytes(C.byref(plan.struct), B, pred_len)
        if nbytes == 0:
            check(-1)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
        pred = torch.empty(B, pred_len, *x.shape[2:], dtype=torch.float32, device=x.device)
        check(lib().sf_rollout_train_fwd_f32(C.byref(plan.struct), x.data_ptr(), pred.data_ptr(), B, pred_len, float(p_drop),
                                             int(seed), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.r, ctx.plan, ctx.ws, ctx.args = r, plan, ws, (B, pred_len, float(p_drop), int(seed), tuple(x.shape))
        ctx.params = params
        return pred

    @staticmethod
    def backward(ctx, d_pred):
        B, pred_len, p_drop, seed, xshape = ctx.args
        r, plan, params = ctx.r, ctx.plan, ctx.params
        d_pred = d_pred.float().contiguous()
        # the weights' gradients, then (learnable tables only) the gradient of the folded table pe_tok [window tokens, d_model]
        tables = learnable_tables(r)
        params = params[:len(params) - len(tables)]
        nw = sum(p.numel() for p in params)
        W, N, d = plan.struct.window_len, plan.struct.num_slots, plan.struct.d_model
        flat = torch.empty(nw + (W * N * d if tables else 0), dtype=torch.float32, device=d_pred.device)
        g, keep = _grad_descriptor(flat, params, len(r.transformer_encoder.layers), nw if tables else None)
        d_x = torch.empty(xshape, dtype=torch.float32, device=d_pred.device) if ctx.needs_input_grad[1] else None
        check(lib().sf_rollout_train_bwd_f32(C.byref(plan.struct), d_pred.data_ptr(), d_x.data_ptr() if d_x is not None else None,
                                             C.byref(g), B, pred_len, p_drop, seed, ctx.ws.data_ptr(), ctx.ws.numel(),
                                             torch.cuda.current_stream().cuda_stream))
        del keep
        ctx.ws = None
        if getattr(r, 'ddp_flat_bucket', False):
            # data-parallel training (config C3): ONE all-reduce of the whole bucket over RCCL / xGMI, averaged
            parallel.allreduce_flat(flat)
        r.last_grad_bucket = flat
        grads = _split(flat, params)
        if tables:
            # pe_tok[t * N + n] = enc_t_pe[t] + enc_slots_pe[n]  (slotformer.py:103-109): sum the folded gradient over slots / frames
            dpe = flat[nw:].view(W, N, d)
            for p in tables:
                grads.append((dpe.sum(1) if p is getattr(r, 'enc_t_pe', None) else dpe.sum(0)).view_as(p))
        return (None, d_x, None, None, None) + tuple(g_ if ctx.needs_input_grad[5 + i] else None for i, g_ in enumerate(grads))


def rollout_with_grad(r, x, pred_len):
    """SlotRollouter.forward under autograd (slotformer.py:85-126).  x [B, history_len, N, C] -> [B, pred_len, N, C]."""
    layer0 = r.transformer_encoder.layers[0]
    p_drop = float(layer0.dropout.p) if r.training else 0.0
    seed = int(torch.randint(0, 2**62, (1, )).item()) if p_drop > 0 else 0
    seed = getattr(r, 'dropout_seed_override', None) or seed
    # (learnable position tables ride along as extra leaves: the kernels read them folded into pe_tok, engine.rollouter_plan, and
    # sf_rollout_train_bwd_f32 returns the folded table's gradient)
    return _Rollout.apply(r, x, pred_len, p_drop, seed, *rollouter_parameters(r), *learnable_tables(r))


# ---------------------------------------------------------------------------------------------------------------------
# Slot Attention (savi.py:36-102) under autograd
# ---------------------------------------------------------------------------------------------------------------------
def slot_attention_parameters(sa):
    """Leaves of a SlotAttention container in the order of sf_slot_attention's pointer fields."""
    return [sa.norm_inputs.weight, sa.norm_inputs.bias, sa.project_k.weight, sa.project_v.weight, sa.project_q[0].weight,
            sa.project_q[0].bias, sa.project_q[1].weight, sa.gru.weight_ih, sa.gru.weight_hh, sa.gru.bias_ih, sa.gru.bias_hh,
            sa.mlp[0].weight, sa.mlp[0].bias, sa.mlp[1].weight, sa.mlp[1].bias, sa.mlp[3].weight, sa.mlp[3].bias]


class _SlotAttention(torch.autograd.Function):
    """slots_out = SlotAttention(inputs, slots) as a single autograd node."""

    @staticmethod
    def forward(ctx, sa, inputs, slots, *params):
        inputs, slots = inputs.detach().float().contiguous(), slots.detach().float().contiguous()
        if not inputs.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
        keep = [p.detach().float().contiguous() for p in params]
        m = sf_slot_attention()
        m.in_features, m.slot_size, m.mlp_hidden, m.num_slots = sa.in_features, sa.slot_size, sa.mlp_hidden_size, slots.shape[1]
        m.eps = float(sa.eps)
        for name, t in zip(_SA_LEAVES, keep):
            setattr(m, name, t.data_ptr())
        B, HW = inputs.shape[:2]
        nbytes = lib().sf_slot_attention_train_workspace_bytes(C.byref(m), B, HW, sa.num_iterations)
        if nbytes == 0:
            check(-1)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=inputs.device)
        out = torch.empty_like(slots)
        check(lib().sf_slot_attention_train_fwd_f32(C.byref(m), inputs.data_ptr(), slots.data_ptr(), B, HW, sa.num_iterations,
                                                    out.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.sa, ctx.m, ctx.keep, ctx.ws, ctx.inputs, ctx.params = sa, m, keep, ws, inputs, params
        return out

    @staticmethod
    def backward(ctx, d_out):
        sa, m, params, inputs = ctx.sa, ctx.m, ctx.params, ctx.inputs
        d_out = d_out.float().contiguous()
        B, HW = inputs.shape[:2]
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=d_out.device)
        g, off = sf_slot_attention_grads(), 0
        for name, p in zip(_SA_LEAVES, params):
            setattr(g, name, flat.data_ptr() + 4 * off)
            off += p.numel()
        d_in = torch.empty_like(inputs) if ctx.needs_input_grad[1] else None
        d_slots = torch.empty_like(d_out)
        check(lib().sf_slot_attention_train_bwd_f32(C.byref(m), inputs.data_ptr(), d_out.data_ptr(),
                                                    d_in.data_ptr() if d_in is not None else None, d_slots.data_ptr(), C.byref(g), B, HW,
                                                    sa.num_iterations, ctx.ws.data_ptr(), ctx.ws.numel(),
                                                    torch.cuda.current_stream().cuda_stream))
        ctx.ws = None
        grads = _split(flat, params)
        return (None, d_in, d_slots if ctx.needs_input_grad[2] else None) + tuple(
            g_ if ctx.needs_input_grad[3 + i] else None for i, g_ in enumerate(grads))


def slot_attention_with_grad(sa, inputs, slots):
    """SlotAttention.forward under autograd (savi.py:56-102): inputs [B,HW,C], slots [B,N,D] -> slots [B,N,D]."""
    return _SlotAttention.apply(sa, inputs, slots, *slot_attention_parameters(sa))


# ---------------------------------------------------------------------------------------------------------------------
# SAVi decoder under autograd (data gradient only: the image term of SlotFormer's loss; the decoder is frozen there)
# ---------------------------------------------------------------------------------------------------------------------
def decoder_parameters(m):
    """Leaves of the spatial-broadcast decoder in bucket order: (weight, bias) per transposed conv, the 1x1 head, the
    position-embedding Linear."""
    n = len(m.dec_channels) - 1
    ps = []
    for i in range(n):
        ps += [m.decoder[i][0].weight, m.decoder[i][0].bias]
    return ps + [m.decoder[n].weight, m.decoder[n].bias, m.decoder_pos_embedding.dense.weight, m.decoder_pos_embedding.dense.bias]


class _Decode(torch.autograd.Function):
    """(recon_combined, recons, masks) = decode(slots); only recon_combined carries a gradient (savi.py:527-538,
    slotformer.py:313-326).  Parameter gradients are produced iff the decoder is not frozen."""

    @staticmethod
    def forward(ctx, m, slots, *params):
        from . import ops
        slots = slots.detach().float().contiguous()
        if not slots.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
        plan = engine.decoder_plan(m, inference=False)
        if not hasattr(plan, 'bwd_w'):
            n = plan.struct.dec_layers
            plan.bwd_keep = [ops.pack_conv_weight(m.decoder[i][0].weight.detach().float().contiguous()) for i in range(n)]
            plan.bwd_w = (C.c_void_p * n)(*[t.data_ptr() for t in plan.bwd_keep])
        F_, N, D = slots.shape
        H = plan.struct.resolution
        nbytes = lib().sf_savi_decode_train_workspace_bytes(C.byref(plan.struct), F_)
        if nbytes == 0:
            check(-1)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=slots.device)
        recon = torch.empty(F_, 3, H, H, device=slots.device, dtype=torch.float32)
        recons = torch.empty(F_, N, 3, H, H, device=slots.device, dtype=torch.float32)
        masks = torch.empty(F_, N, 1, H, H, device=slots.device, dtype=torch.float32)
        check(lib().sf_savi_decode_train_fwd_f32(C.byref(plan.struct), slots.data_ptr(), recon.data_ptr(), recons.data_ptr(),
                                                 masks.data_ptr(), F_, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.m, ctx.plan, ctx.ws, ctx.shape, ctx.params = m, plan, ws, (F_, N, D), params
        ctx.mark_non_differentiable(recons, masks)
        return recon, recons, masks

    @staticmethod
    def backward(ctx, d_recon, _d_recons, _d_masks):
        F_, N, D = ctx.shape
        params, m = ctx.params, ctx.m
        d_recon = d_recon.float().contiguous()
        d_slots = torch.empty(F_, N, D, device=d_recon.device, dtype=torch.float32)
        g, grid, flat = None, None, None
        if params:
            flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=d_recon.device)
            ptrs, off = [], 0
            for p in params:
                ptrs.append(flat.data_ptr() + 4 * off)
                off += p.numel()
            g = sf_savi_decoder_grads()
            n = ctx.plan.struct.dec_layers
            for i in range(n):
                g.deconv_w[i], g.deconv_b[i] = ptrs[2 * i], ptrs[2 * i + 1]
            g.out_w, g.out_b, g.pos_w, g.pos_b = ptrs[2 * n:2 * n + 4]
            grid = m.decoder_pos_embedding.grid.detach().float().reshape(-1, 4).contiguous()
        check(lib().sf_savi_decode_train_bwd_f32(C.byref(ctx.plan.struct), ctx.plan.bwd_w, d_recon.data_ptr(), d_slots.data_ptr(),
                                                 grid.data_ptr() if grid is not None else None, C.byref(g) if g is not None else None,
                                                 F_, ctx.ws.data_ptr(), ctx.ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.ws = None
        grads = ()
        if params:
            grads = tuple(g_ if ctx.needs_input_grad[2 + i] else None for i, g_ in enumerate(_split(flat, params)))
        return (None, d_slots if ctx.needs_input_grad[1] else None) + grads


def decode_with_grad(m, slots):
    """StoSAVi.decode (savi.py:504-525) under autograd: d(recon_combined)/d(slots), plus the decoder's own parameter
    gradients unless it is frozen (as in SlotFormer, slotformer.py:203-210)."""
    params = decoder_parameters(m)
    if not any(p.requires_grad for p in params):
        params = []
    return _Decode.apply(m, slots, *params)


# ---------------------------------------------------------------------------------------------------------------------
# SAVi image encoder (conv stack + position embedding + per-pixel MLP, savi.py:220-250,367-377) under autograd
# ---------------------------------------------------------------------------------------------------------------------
def features_parameters(m):
    """Leaves of a StoSAVi / STEVE container that the encoder node differentiates, in bucket order."""
    ps = []
    for i in range(len(m.enc_channels) - 1):
        ps += [m.encoder[i][0].weight, m.encoder[i][0].bias]
    pe, eo = m.encoder_pos_embedding, m.encoder_out_layer
    return ps + [pe.dense.weight, pe.dense.bias, eo[0].weight, eo[0].bias, eo[1].weight, eo[1].bias, eo[3].weight, eo[3].bias]


class _Features(torch.autograd.Function):
    """encoder_out [F, 64*64, C_out] = encoder_out_layer(flatten(encoder(img) + pos)) as one node (no image gradient)."""

    @staticmethod
    def forward(ctx, m, img, *params):
        img = img.detach().float().contiguous()
        if not img.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
        keep = [p.detach().float().contiguous() for p in params]
        n = len(m.enc_channels) - 1
        s = sf_savi_features()
        s.resolution, s.layers, s.channels, s.ks = m.resolution[0], n, m.enc_channels[1], m.enc_ks
        s.hidden, s.out_channels = m.encoder_out_layer[1].out_features, m.encoder_out_layer[3].out_features
        if any(c != s.channels for c in m.enc_channels[1:]):
            raise NotImplementedError('slotformer_amd: encoder training needs equal conv widths (the reference uses 64 everywhere)')
        for i in range(n):
            s.conv_w[i], s.conv_b[i] = keep[2 * i].data_ptr(), keep[2 * i + 1].data_ptr()
        grid = m.encoder_pos_embedding.grid.detach().float().reshape(-1, 4).contiguous()
        keep.append(grid)
        s.pos_grid = grid.data_ptr()
        for name, t in zip(('pos_w', 'pos_b', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b'), keep[2 * n:2 * n + 8]):
            setattr(s, name, t.data_ptr())
        F_ = img.shape[0]
        nbytes = lib().sf_savi_features_train_workspace_bytes(C.byref(s), F_)
        if nbytes == 0:
            check(-1)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=img.device)
        out = torch.empty(F_, 64 * 64, s.out_channels, dtype=torch.float32, device=img.device)
        check(lib().sf_savi_features_train_fwd_f32(C.byref(s), img.data_ptr(), img[0].numel(), F_, out.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.m, ctx.s, ctx.keep, ctx.ws, ctx.img, ctx.params = m, s, keep, ws, img, params
        return out

    @staticmethod
    def backward(ctx, d_out):
        s, params, img = ctx.s, ctx.params, ctx.img
        d_out = d_out.float().contiguous()
        flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=d_out.device)
        ptrs, off = [], 0
        for p in params:
            ptrs.append(flat.data_ptr() + 4 * off)
            off += p.numel()
        g = sf_savi_features_grads()
        n = s.layers
        for i in range(n):
            g.conv_w[i], g.conv_b[i] = ptrs[2 * i], ptrs[2 * i + 1]
        for name, ptr in zip(('pos_w', 'pos_b', 'ln_g', 'ln_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b'), ptrs[2 * n:]):
            setattr(g, name, ptr)
        check(lib().sf_savi_features_train_bwd_f32(C.byref(s), img.data_ptr(), img[0].numel(), d_out.data_ptr(), C.byref(g), img.shape[0],
                                                   ctx.ws.data_ptr(), ctx.ws.numel(), torch.cuda.current_stream().cuda_stream))
        ctx.ws = None
        grads = _split(flat, params)
        return (None, None) + tuple(g_ if ctx.needs_input_grad[2 + i] else None for i, g_ in enumerate(grads))


def features_with_grad(m, img):
    """img [F, 3, H, W] -> encoder_out [F, 64*64, enc_out_channels] under autograd (savi.py:367-377)."""
    return _Features.apply(m, img, *features_parameters(m))


# ---------------------------------------------------------------------------------------------------------------------
# differentiable nn.Linear / nn.LayerNorm on the HIP library, for the slot-level layers (predictor, kernel distribution)
# ---------------------------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) with sf_linear_f32 / sf_linear_bwd_f32."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        from . import ops
        xd, wd = x.detach().float().contiguous(), weight.detach().float().contiguous()
        y = ops.linear(xd, wd, bias.detach().float().contiguous() if bias is not None else None, relu=relu)
        ctx.save_for_backward(xd, wd, y if relu else None)
        ctx.relu, ctx.has_bias = bool(relu), bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.float().contiguous().clone() if ctx.relu else dy.float().contiguous()
        N, K = w.shape
        M = x.numel() // K
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=w.device) if ctx.has_bias else None
        nb = lib().sf_linear_bwd_workspace_bytes(M, N, K)
        ws = torch.empty(nb, dtype=torch.uint8, device=w.device)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        check(lib().sf_linear_bwd_f32(p(x), p(w), p(y), p(dy), p(dx), p(dw), p(db), M, N, K, int(ctx.relu), ws.data_ptr(), nb,
                                      torch.cuda.current_stream().cuda_stream))
        return dx, dw, db, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        from . import ops
        xd, g = x.detach().float().contiguous(), gamma.detach().float().contiguous()
        ctx.save_for_backward(xd, g)
        ctx.eps = float(eps)
        return ops.layernorm(xd, g, beta.detach().float().contiguous(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, g = ctx.saved_tensors
        dy = dy.float().contiguous()
        D = x.shape[-1]
        dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(g)
        nb = lib().sf_layernorm_bwd_workspace_bytes(D)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        check(lib().sf_layernorm_bwd_f32(x.data_ptr(), dy.data_ptr(), g.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                         x.numel() // D, D, ctx.eps, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream))
        return dx, dg, db, None


class _MHA(torch.autograd.Function):
    """Attention core of nn.MultiheadAttention (batch_first self-attention) on packed q|k|v rows."""

    @staticmethod
    def forward(ctx, qkv, B, L, d, heads, p, seed):
        qkv = qkv.detach().float().contiguous()
        out = torch.empty(B * L, d, dtype=torch.float32, device=qkv.device)
        check(lib().sf_mha_train_fwd_f32(qkv.data_ptr(), out.data_ptr(), B, L, d, heads, float(p), int(seed),
                                         torch.cuda.current_stream().cuda_stream))
        ctx.save_for_backward(qkv)
        ctx.args = (B, L, d, heads, float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, d_out):
        qkv, = ctx.saved_tensors
        B, L, d, heads, p, seed = ctx.args
        d_out = d_out.float().contiguous()
        dqkv = torch.empty_like(qkv)
        check(lib().sf_mha_train_bwd_f32(qkv.data_ptr(), d_out.data_ptr(), dqkv.data_ptr(), B, L, d, heads, p, seed,
                                         torch.cuda.current_stream().cuda_stream))
        return dqkv, None, None, None, None, None, None


class _Dropout(torch.autograd.Function):
    """nn.Dropout in train mode with the library's hashed masks (regenerated in the backward pass)."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.detach().float().contiguous()
        y = torch.empty_like(x)
        check(lib().sf_dropout_f32(x.data_ptr(), None, y.data_ptr(), x.numel(), float(p), int(seed), torch.cuda.current_stream().cuda_stream))
        ctx.args = (float(p), int(seed))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.float().contiguous()
        dx = torch.empty_like(dy)
        check(lib().sf_dropout_f32(dy.data_ptr(), None, dx.data_ptr(), dy.numel(), ctx.args[0], ctx.args[1], torch.cuda.current_stream().cuda_stream))
        return dx, None, None


def _new_seed():
    return int(torch.randint(0, 2**62, (1, )).item())


def dropout(x, p, training):
    return _Dropout.apply(x, p, _new_seed()) if (training and p > 0) else x


def transformer_encoder_layer(x, layer):
    """nn.TransformerEncoderLayer (pre-LN, relu, batch_first; predictor.py:33-38) on x [B, L, d] as a chain of HIP-backed
    autograd nodes, dropout included when the layer is in train() mode."""
    if not layer.norm_first:
        raise NotImplementedError('slotformer_amd: training needs norm_first Transformer layers (all reference configurations)')
    B, L, d = x.shape
    tr, att = layer.training, layer.self_attn
    h = layer_norm(x, layer.norm1)
    qkv = _Linear.apply(h, att.in_proj_weight, att.in_proj_bias, False)
    p_att = float(att.dropout) if tr else 0.0
    ctx = _MHA.apply(qkv.reshape(B * L, 3 * d), B, L, d, att.num_heads, p_att, _new_seed() if p_att > 0 else 0).view(B, L, d)
    x = x + dropout(linear(ctx, att.out_proj), layer.dropout1.p, tr)
    h = layer_norm(x, layer.norm2)
    h = dropout(linear(h, layer.linear1, relu=True), layer.dropout.p, tr)
    return x + dropout(linear(h, layer.linear2), layer.dropout2.p, tr)


def lstm_step(x, state, rnn):
    """One step of a 1-layer nn.LSTM on x [R, C] with state (h, c) [R, H] or None (predictor.py:116-117); the two gate
    projections are HIP-backed nodes, the gate nonlinearities are elementwise glue."""
    R = x.shape[0]
    H = rnn.hidden_size
    if state is None:
        h = torch.zeros(R, H, dtype=torch.float32, device=x.device)
        c = torch.zeros_like(h)
    else:
        h, c = state
    gates = _Linear.apply(x, rnn.weight_ih_l0, rnn.bias_ih_l0, False) + _Linear.apply(h, rnn.weight_hh_l0, rnn.bias_hh_l0, False)
    i, f, g, o = gates.chunk(4, -1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, (h2, c2)


class _SlateAttention(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd) [causal]) v of the STEVE decoder (steve_transformer.py:12-55) on contiguous q [B,Lq,d],
    k, v [B,Lk,d], with optional dropout on the attention weights (p > 0: the training forward, which also keeps the row
    log-sum-exp; p == 0: the inference kernel).  Backward: the flash-style sf_slate_attention_train_bwd_f32."""

    @staticmethod
    def forward(ctx, q, k, v, heads, causal, p, seed):
        from . import ops
        q, k, v = (t.detach().float().contiguous() for t in (q, k, v))
        B, Lq, d = q.shape
        Lk = k.shape[1]
        lse = None
        if p > 0:
            out = torch.empty_like(q)
            lse = torch.empty(B, heads, Lq, dtype=torch.float32, device=q.device)
            check(lib().sf_slate_attention_train_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), d, d, d, d,
                                                         Lq * d, Lk * d, Lk * d, Lq * d, B, Lq, Lk, heads, d // heads, int(causal), float(p),
                                                         int(seed), torch.cuda.current_stream().cuda_stream))
        else:
            out = ops.slate_attention(q, k, v, heads, causal)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.args = (heads, bool(causal), float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, k, v, out, lse = ctx.saved_tensors
        heads, causal, p, seed = ctx.args
        B, Lq, d = q.shape
        Lk = k.shape[1]
        d_out = d_out.float().contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        nb = lib().sf_slate_attention_bwd_workspace_bytes(B, Lq, heads)
        ws = torch.empty(nb, dtype=torch.uint8, device=q.device)
        check(lib().sf_slate_attention_train_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), d_out.data_ptr(),
                                                     lse.data_ptr() if lse is not None else None, dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                     d, d, d, d, Lq * d, Lk * d, Lk * d, Lq * d, B, Lq, Lk, heads, d // heads, int(causal), p,
                                                     seed, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream))
        return dq, dk, dv, None, None, None, None


class _SelfAttention(torch.autograd.Function):
    """Self-attention of a STEVE decoder block from its input: q|k|v = x [Wq;Wk;Wv]^T in ONE GEMM into a packed [B,L,3d]
    buffer that the attention kernels read in place (column offsets 0, d, 2d), and whose gradient the attention backward
    writes in place -- no per-projection GEMMs, no copies of q, k, v or of their gradients."""

    @staticmethod
    def forward(ctx, x, wq, wk, wv, heads, causal, p, seed):
        from . import ops
        x = x.detach().float().contiguous()
        B, L, d = x.shape
        wcat = torch.cat([w.detach().float() for w in (wq, wk, wv)], 0).contiguous()
        qkv = ops.linear(x, wcat)                                   # [B, L, 3d]
        lse = None
        if p > 0:
            out = torch.empty(B, L, d, dtype=torch.float32, device=x.device)
            lse = torch.empty(B, heads, L, dtype=torch.float32, device=x.device)
            base = qkv.data_ptr()
            check(lib().sf_slate_attention_train_fwd_f32(base, base + 4 * d, base + 8 * d, out.data_ptr(), lse.data_ptr(), 3 * d, 3 * d,
                                                         3 * d, d, L * 3 * d, L * 3 * d, L * 3 * d, L * d, B, L, L, heads, d // heads,
                                                         int(causal), float(p), int(seed), torch.cuda.current_stream().cuda_stream))
        else:
            out = ops.slate_attention(qkv, qkv, qkv, heads, causal, 0, d, 2 * d, d_model=d)
        ctx.save_for_backward(x, wcat, qkv, out, lse)
        ctx.args = (heads, bool(causal), float(p), int(seed))
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, wcat, qkv, out, lse = ctx.saved_tensors
        heads, causal, p, seed = ctx.args
        B, L, d = x.shape
        d_out = d_out.float().contiguous()
        dqkv = torch.empty_like(qkv)
        st = torch.cuda.current_stream().cuda_stream
        nb = lib().sf_slate_attention_bwd_workspace_bytes(B, L, heads)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        b0, g0 = qkv.data_ptr(), dqkv.data_ptr()
        check(lib().sf_slate_attention_train_bwd_f32(b0, b0 + 4 * d, b0 + 8 * d, out.data_ptr(), d_out.data_ptr(),
                                                     lse.data_ptr() if lse is not None else None, g0, g0 + 4 * d, g0 + 8 * d, 3 * d, 3 * d,
                                                     3 * d, d, L * 3 * d, L * 3 * d, L * 3 * d, L * d, B, L, L, heads, d // heads,
                                                     int(causal), p, seed, ws.data_ptr(), nb, st))
        dx = torch.empty_like(x)
        dw = torch.empty_like(wcat)
        M = B * L
        nb = lib().sf_linear_bwd_workspace_bytes(M, 3 * d, d)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        check(lib().sf_linear_bwd_f32(x.data_ptr(), wcat.data_ptr(), None, dqkv.data_ptr(), dx.data_ptr(), dw.data_ptr(), None, M, 3 * d, d,
                                      0, ws.data_ptr(), nb, st))
        return (dx, ) + tuple(dw.split(d, 0)) + (None, None, None, None)


class _Embed(torch.autograd.Function):
    """tok_emb[idx] + pos[:L] (steve_transformer.py:58-74,286-289): HIP gather forward; the backward scatters the row
    gradients back (index_add on the [V+1, d] table, batch sum for the position table)."""

    @staticmethod
    def forward(ctx, idx, tok_emb, pos):
        from . import ops
        ctx.save_for_backward(idx)
        ctx.shapes = (tuple(tok_emb.shape), tuple(pos.shape))
        return ops.embed_tokens(idx, tok_emb.detach().float().contiguous(), pos.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dx):
        idx, = ctx.saved_tensors
        (V, d), pshape = ctx.shapes
        dx = dx.float().contiguous()
        d_tok = torch.zeros(V, d, dtype=torch.float32, device=dx.device).index_add_(0, idx.reshape(-1), dx.reshape(-1, d))
        d_pos = torch.zeros(pshape, dtype=torch.float32, device=dx.device)
        d_pos[:dx.shape[1]] = dx.sum(0)
        return None, d_tok, d_pos


class _TokenCrossEntropy(torch.autograd.Function):
    """mean cross-entropy of logits [R,V] against int64 targets [R] (steve.py:341-344): sf_cross_entropy_f32 forward,
    (softmax - onehot) / R backward on sf_softmax_rows_f32."""

    @staticmethod
    def forward(ctx, logits, target):
        from . import ops
        logits = logits.detach().float().contiguous()
        ctx.save_for_backward(logits, target)
        return ops.cross_entropy(logits, target)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        logits, target = ctx.saved_tensors
        R, V = logits.shape
        if V > 16384:
            p = ops.softmax_rows(logits)
            p.scatter_add_(1, target.view(-1, 1), torch.full((target.numel(), 1), -1.0, device=p.device))
            return p * (g / R), None
        dx = torch.empty_like(logits)
        gs = g.detach().float().reshape(1).contiguous()
        check(lib().sf_cross_entropy_bwd_f32(logits.data_ptr(), target.data_ptr(), gs.data_ptr(), dx.data_ptr(), R, V,
                                             torch.cuda.current_stream().cuda_stream))
        return dx, None


def slate_decoder_forward(dec, slots, idx):
    """STEVETransformerDecoder.forward (steve_transformer.py:275-303) under autograd: slots [B,N,d], idx int64 [B,t] ->
    logits [B,1+t,V], as a chain of HIP-backed nodes; every dropout of the reference modules (embedding, attention weights,
    attention output, FFN) is active iff the decoder is in train() mode."""
    tr = dec.training
    B, T = idx.shape
    H = dec.n_head

    def attend(q, k, v, mha, causal):
        p = float(mha.attn_dropout.p) if tr else 0.0
        return _SlateAttention.apply(q, k, v, H, causal, p, _new_seed() if p > 0 else 0)
    mem = linear(slots.float().contiguous(), dec.in_proj)
    bos = torch.full((B, 1), dec.vocab_size, dtype=torch.int64, device=idx.device)
    tokens = torch.cat([bos, idx.to(torch.int64)], 1).contiguous()
    x = dropout(_Embed.apply(tokens, dec.tok_emb.weight, dec.pos_emb.pe[0]), dec.pos_emb.dropout.p, tr)
    for blk in dec.tf_dec.blocks:
        sa, ca = blk.self_attn, blk.encoder_decoder_attn
        y = layer_norm(x, blk.self_attn_layer_norm)
        if blk.is_first:   # the first block normalises its input in place (steve_transformer.py:186-190)
            x = y
        p_sa = float(sa.attn_dropout.p) if tr else 0.0
        att = _SelfAttention.apply(y, sa.proj_q.weight, sa.proj_k.weight, sa.proj_v.weight, H, True, p_sa, _new_seed() if p_sa > 0 else 0)
        x = x + dropout(linear(att, sa.proj_o), sa.output_dropout.p, tr)
        y = layer_norm(x, blk.encoder_decoder_attn_layer_norm)
        att = attend(linear(y, ca.proj_q), *linear_cat(mem, ca.proj_k, ca.proj_v), ca, False)
        x = x + dropout(linear(att, ca.proj_o), ca.output_dropout.p, tr)
        y = layer_norm(x, blk.ffn_layer_norm)
        x = x + dropout(linear(linear(y, blk.ffn[0], relu=True), blk.ffn[2]), blk.ffn[3].p, tr)
    return linear(layer_norm(x, dec.tf_dec.layer_norm), dec.head)


def token_cross_entropy(logits, target):
    return _TokenCrossEntropy.apply(logits, target.to(torch.int64).contiguous())


class _LinearCat(torch.autograd.Function):
    """[x W1^T | x W2^T | ...] for bias-free projections that share their input (q|k|v of the STEVE decoder's attention,
    steve_transformer.py:30-40): ONE GEMM on the concatenated weight, one weight-gradient contraction, one data-gradient GEMM,
    instead of one of each per projection."""

    @staticmethod
    def forward(ctx, x, *weights):
        from . import ops
        xd = x.detach().float().contiguous()
        wcat = torch.cat([w.detach().float() for w in weights], 0).contiguous()
        ctx.save_for_backward(xd, wcat)
        ctx.sizes = [w.shape[0] for w in weights]
        return ops.linear(xd, wcat)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.float().contiguous()
        N, K = w.shape
        M = x.numel() // K
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        nb = lib().sf_linear_bwd_workspace_bytes(M, N, K)
        ws = torch.empty(nb, dtype=torch.uint8, device=w.device)
        check(lib().sf_linear_bwd_f32(x.data_ptr(), w.data_ptr(), None, dy.data_ptr(), dx.data_ptr() if dx is not None else None,
                                      dw.data_ptr(), None, M, N, K, 0, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream))
        return (dx, ) + tuple(dw.split(ctx.sizes, 0))


def linear_cat(x, *layers):
    """x projected by several bias-free nn.Linear layers at once; returns the tuple of outputs (column slices)."""
    out = _LinearCat.apply(x, *[l_.weight for l_ in layers])
    return out.split([l_.weight.shape[0] for l_ in layers], -1)


def linear(x, layer, relu=False):
    """nn.Linear `layer` applied to x under autograd, on the HIP kernels."""
    return _Linear.apply(x, layer.weight, layer.bias, relu)


def layer_norm(x, layer):
    return _LayerNorm.apply(x, layer.weight, layer.bias, layer.eps)


# ---- dVAE nodes (dVAE.py:113-139, steve_utils.py:26-41, 100-126) ------------------------------------------------------
class _GroupNorm1(torch.autograd.Function):
    """Conv2dBlock's GroupNorm(1 group) + ReLU (+ the PixelShuffle(2) that follows it) on NHWC maps:
    sf_groupnorm1_nhwc_f32 / sf_groupnorm1_nhwc_bwd_f32 (statistics and ReLU gate recomputed from the saved input)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, relu, pixel_shuffle):
        from . import ops
        xd, g, b = x.detach().float().contiguous(), gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        ctx.save_for_backward(xd, g, b)
        ctx.cfg = (float(eps), int(bool(relu)), int(pixel_shuffle))
        return ops.groupnorm1_nhwc(xd, g, b, eps=eps, relu=relu, pixel_shuffle=pixel_shuffle)

    @staticmethod
    def backward(ctx, dy):
        x, g, b = ctx.saved_tensors
        eps, relu, r = ctx.cfg
        dy = dy.float().contiguous()
        F_, H, W, C_ = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(b)
        nb = lib().sf_groupnorm1_bwd_workspace_bytes(F_, H, W, C_)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        check(lib().sf_groupnorm1_nhwc_bwd_f32(x.data_ptr(), g.data_ptr(), b.data_ptr(), dy.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                               db.data_ptr(), F_, H, W, C_, eps, relu, r, ws.data_ptr(), nb,
                                               torch.cuda.current_stream().cuda_stream))
        return dx, dg, db, None, None, None


def groupnorm1(x, gamma, beta, eps=1e-5, relu=True, pixel_shuffle=1):
    return _GroupNorm1.apply(x, gamma, beta, eps, relu, pixel_shuffle)


class _SoftmaxRows(torch.autograd.Function):
    """softmax((x + noise) * scale) over the last dim; with Gumbel noise and scale = 1/tau the relaxed one-hot sample of
    steve_utils.py:26-41 (the log-partition shift of log_softmax cancels in the softmax).  The noise is either the tensor
    `add` or, with `seed`, generated inside the kernel (sf_gumbel_softmax_rows_f32) and never stored: the backward only
    needs the output."""

    @staticmethod
    def forward(ctx, x, add, scale, seed):
        from . import ops
        xd = x.detach().float().contiguous()
        if seed is not None and add is None:
            y = ops.gumbel_softmax_rows(xd, seed, scale)
        else:
            y = ops.softmax_rows(xd, None if add is None else add.detach().float().contiguous(), scale)
        ctx.save_for_backward(y)
        ctx.scale = float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = dy.float().contiguous()
        V = y.shape[-1]
        dx = torch.empty_like(y)
        check(lib().sf_softmax_rows_bwd_f32(y.data_ptr(), dy.data_ptr(), ctx.scale, dx.data_ptr(), y.numel() // V, V,
                                            torch.cuda.current_stream().cuda_stream))
        return dx, None, None, None


def gumbel_softmax(logits, gumbels=None, tau=1., hard=False, seed=None):
    """steve_utils.py:26-41 on the last dim: relaxed sample, or the straight-through one-hot.  The Gumbel noise is `gumbels`
    when given, else drawn inside the kernel from `seed` (default: a fresh seed from torch's CPU generator)."""
    from . import ops
    if gumbels is None and seed is None:
        seed = _new_seed()
    y_soft = _SoftmaxRows.apply(logits, gumbels, 1. / tau, seed)
    if not hard:
        return y_soft
    idx = ops.argmax_rows(y_soft.detach())
    y_hard = torch.zeros_like(y_soft).scatter_(-1, idx.unsqueeze(-1), 1.)
    return y_hard - y_soft.detach() + y_soft


def gumbel_noise(seed, numel):
    """Host restatement of the in-kernel Gumbel noise (steve_decoder.hip: sf_gumbel) for tests: float32 [numel]."""
    import numpy as np

    def mix(h):
        h = h.astype(np.uint32)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x7feb352d)).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h * np.uint32(0x846ca68b)).astype(np.uint32)
        h ^= h >> np.uint32(16)
        return h

    with np.errstate(over='ignore'):
        inner = mix(np.array([np.uint32((seed >> 32) & 0xffffffff) + np.uint32(0x9e3779b9)], dtype=np.uint32))
        sseed = mix(np.array([np.uint32(seed & 0xffffffff)], dtype=np.uint32) ^ inner)[0]
        h = mix(np.arange(numel, dtype=np.uint32) ^ sseed)
    u = ((h >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.1920929e-7)
    return torch.from_numpy(-np.log(-np.log(u)))


def linear_weight(x, weight, bias=None, relu=False):
    """act(x W^T + b) for a bare [N, K] weight under autograd.  The backward contractions work on multiples of 64, so other
    widths (the dVAE's 48-wide patch vectors, its 3-channel output) are zero-padded up and the result sliced back."""
    N, K = weight.shape
    Np, Kp = -(-N // 64) * 64, -(-K // 64) * 64
    if Kp != K:
        x, weight = tnf.pad(x, (0, Kp - K)), tnf.pad(weight, (0, Kp - K))
    if Np != N:
        weight = tnf.pad(weight, (0, 0, 0, Np - N))
        bias = tnf.pad(bias, (0, Np - N)) if bias is not None else None
    y = _Linear.apply(x, weight, bias, relu)
    return y[..., :N] if Np != N else y


def conv3x3_nhwc(x, weight):
    """Bias-free 3x3 / stride-1 / pad-1 convolution of an NHWC map under autograd (the two 3x3 Conv2dBlocks of the dVAE decoder,
    dVAE.py:39,45) as ONE GEMM over the nine shifted copies of the map: the copies and their adjoint (shift-and-add) are data
    movement, the contraction and both of its gradients run on the HIP GEMM kernels."""
    F_, H, W, C_ = x.shape
    xp = tnf.pad(x, (0, 0, 1, 1, 1, 1))
    cols = torch.cat([xp[:, ky:ky + H, kx:kx + W] for ky in range(3) for kx in range(3)], -1)   # [F,H,W,9C], (ky, kx, c) order
    w2 = weight.permute(0, 2, 3, 1).reshape(weight.shape[0], 9 * C_)
    return linear_weight(cols, w2)


class FlatAdam:
    """Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay -- the reference's optimiser) over ONE
    flat bucket.  The parameters are re-pointed at views of a single fp32 tensor (their values are kept), gradients are
    gathered into a matching flat tensor, and a step is one `sf_adam_flat_f32` launch -- plus, under data parallelism
    (`allreduce=True`), one RCCL all-reduce of that gradient bucket first.  `lr` may be changed between steps (schedules)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, allreduce=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError('slotformer_amd: FlatAdam needs parameters on a HIP device; there is no CPU fallback')
        self.lr, self.betas, self.eps, self.allreduce, self.steps = float(lr), betas, float(eps), allreduce, 0
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)   # same values, now a view of the bucket
            off += k
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.grad)
        self.exp_avg_sq = torch.zeros_like(self.grad)

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def step(self):
        off = 0
        for p in self.params:
            k = p.numel()
            if p.grad is None:
                self.grad[off:off + k].zero_()
            else:
                self.grad[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        if self.allreduce:
            parallel.allreduce_flat(self.grad)
        self.steps += 1
        check(lib().sf_adam_flat_f32(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                     self.flat.numel(), self.steps, self.lr, float(self.betas[0]), float(self.betas[1]), self.eps,
                                     torch.cuda.current_stream().cuda_stream))
        # The kernel wrote the parameters through a raw pointer: tell torch (and, through `_version`, every plan cache
        # in engine.py that is keyed on (data_ptr, _version)) that their contents changed -- otherwise derived / packed
        # weight copies built before this step would keep being served.
        for p in self.params:
            torch._C._increment_version(p)


class amp_bf16:
    """Context manager for the AMP-bf16 policy of the training path: inside it the GEMM / convolution / weight-gradient
    cores contract single-pass bf16 operands with f32 accumulation (library precision mode 2); parameters, activations,
    gradients and the optimizer state stay fp32.  The counterpart of the reference's `--fp16` autocast (train.py:84)."""

    def __enter__(self):
        self.old = lib().sf_get_precision()
        check(lib().sf_set_precision(2))
        return self

    def __exit__(self, *exc):
        check(lib().sf_set_precision(self.old))
        return False


def dropout_keep_mask(seed, step, layer, site, numel, p):
    """Host restatement of the library's dropout mask (rollout_train.hip: sf_keep / site_seed) for tests and tools:
    bool [numel], True = kept.  site: 0 attention weights, 1 attention output, 2 FFN hidden, 3 FFN output."""
    import numpy as np

    def mix(h):
        h = h.astype(np.uint32)
        h ^= h >> np.uint32(16)
        h = (h * np.uint32(0x7feb352d)).astype(np.uint32)
        h ^= h >> np.uint32(15)
        h = (h * np.uint32(0x846ca68b)).astype(np.uint32)
        h ^= h >> np.uint32(16)
        return h

    with np.errstate(over='ignore'):
        tag = np.uint32((step * 64 + layer) * 4 + site)
        inner = mix(np.array([np.uint32((seed >> 32) & 0xffffffff) + np.uint32(0x9e3779b9) * (tag + np.uint32(1))], dtype=np.uint32))
        sseed = mix(np.array([np.uint32(seed & 0xffffffff)], dtype=np.uint32) ^ inner)[0]
        idx = np.arange(numel, dtype=np.uint32)
        u = mix(idx ^ sseed) >> np.uint32(8)
    return u >= np.uint32(int(float(np.float32(p)) * 16777216.0))
