"""Host-side glue between the nn.Module parameter containers and libslotformer_hip.

Builds the ctypes model descriptors (sf_savi_encoder / sf_rollouter) from module parameters,
packs the few derived constants (conv weights OHWI, [Wk;Wv], position tables) with the
library's own kernels, owns the per-device workspace, and launches the whole-path engines on
torch's current stream.  PyTorch is used for device memory and streams only.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import lib, check, sf_tfm_layer, sf_rollouter, sf_savi_encoder, sf_savi_decoder
from . import ops

_WORKSPACES = {}


def workspace(device, nbytes, slot=0):
    """Grow-only per-(device, slot) scratch buffer.  Calls that may run concurrently on different
    streams must use different slots."""
    key = (device.type, device.index, slot)
    cur = _WORKSPACES.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = cur
    return cur


def release_workspaces(prefix):
    """Drop every workspace whose slot key starts with `prefix` (a tuple): an owner that captured raw pointers into its
    workspaces (pipeline graphs) uses private slot keys and frees them when it is closed."""
    n = len(prefix)
    for key in [k for k in _WORKSPACES if isinstance(k[2], tuple) and len(k[2]) > 1 and isinstance(k[2][1], tuple) and k[2][1][:n] == prefix]:
        del _WORKSPACES[key]


def kernel_noise(m, noise, B, T, device):
    """The stochastic-kernel noise an encode call of `m` uses: None when the model samples nothing -- no kernel_dist layer,
    or `kld_method == 'none'`, where `_sample_dist` returns the mean (savi.py:355-365; the OBJ3D / PHYRE configurations) --
    else the caller's tensor, else fresh eps ~ N(0,1) per frame as the reference draws it.  Every caller of savi_encode that
    does not go through StoSAVi.encode (pipeline, harness) must use this, or an untrained log-variance head corrupts the slots."""
    if getattr(m, 'kernel_dist_layer', None) is None or getattr(m, 'kld_method', None) == 'none':
        return None
    if noise is not None:
        return noise
    return torch.randn(B, T, m.num_slots, m.slot_size, device=device)


def _require_inference(module, *tensors):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            'slotformer_amd: this call is inference-only on the HIP path (the training nodes of row N1 live in train.py); '
            'wrap the call in torch.no_grad()')
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')


class _Plan:
    """ctypes descriptor + the tensors it points to."""

    def __init__(self):
        self.keep = []
        self.struct = None
        self.sig = None

    # plans hold ctypes structures with raw pointers: a copied / pickled module starts without one and rebuilds it
    def __deepcopy__(self, memo):
        return _Plan()

    def __reduce__(self):
        return (_Plan, ())

    def dp(self, t):
        if t is None:
            return None
        t = t.detach()
        if t.dtype != torch.float32 or not t.is_contiguous():
            t = t.float().contiguous()
        self.keep.append(t)
        return t.data_ptr()


_PLAN_ATTRS = ('_sf_plan', '_sf_train_plan', '_sf_dec_plan', '_sf_dec_train_plan', '_plan', '_cat', '_catw', '_slate_plan')


def invalidate(module):
    """Drop every cached plan (packed / derived weight copies) below `module`.  The caches are keyed on
    (data_ptr, _version) of the parameters, which follows optimizer steps and ordinary in-place ops; writes that bypass
    torch's version counter (`p.data.copy_()`, raw-pointer kernels) need this call -- or `torch._C._increment_version(p)`."""
    for m in module.modules():
        for a in _PLAN_ATTRS:
            if a in m.__dict__:
                del m.__dict__[a]


def _signature(module):
    return tuple((t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers()))


def _tfm_layers(plan, encoder, pack_ffn=False):
    """nn.TransformerEncoder -> host array of sf_tfm_layer.  pack_ffn: also build the fragment-ordered split-bf16
    copies of linear1 / linear2 that the two-launch rollout layer reads (d_model 256, ffn 1024 only)."""
    n = len(encoder.layers)
    arr = (sf_tfm_layer * n)()
    for i, l in enumerate(encoder.layers):
        a = arr[i]
        d, ffn = l.linear1.in_features, l.linear1.out_features
        if pack_ffn and d == 256 and ffn == 1024 and l.linear1.weight.is_cuda:
            nb = lib().sf_ffn_packed_bytes(d, ffn)
            p1 = torch.empty(nb, dtype=torch.uint8, device=l.linear1.weight.device)
            p2 = torch.empty(nb, dtype=torch.uint8, device=l.linear1.weight.device)
            check(lib().sf_pack_ffn_weights(plan.dp(l.linear1.weight), plan.dp(l.linear2.weight), p1.data_ptr(),
                                            p2.data_ptr(), d, ffn, torch.cuda.current_stream().cuda_stream))
            plan.keep += [p1, p2]
            a.lin1_packed, a.lin2_packed = p1.data_ptr(), p2.data_ptr()
            if l.self_attn.num_heads == 8:
                q1 = torch.empty(lib().sf_attn_packed_bytes(d, 0), dtype=torch.uint8, device=p1.device)
                q2 = torch.empty(lib().sf_attn_packed_bytes(d, 1), dtype=torch.uint8, device=p1.device)
                check(lib().sf_pack_attn_weights(plan.dp(l.self_attn.in_proj_weight), plan.dp(l.self_attn.out_proj.weight),
                                                 q1.data_ptr(), q2.data_ptr(), d, 8, torch.cuda.current_stream().cuda_stream))
                plan.keep += [q1, q2]
                a.attn_in_packed, a.attn_out_packed = q1.data_ptr(), q2.data_ptr()
        a.norm1_g, a.norm1_b = plan.dp(l.norm1.weight), plan.dp(l.norm1.bias)
        a.in_proj_w, a.in_proj_b = plan.dp(l.self_attn.in_proj_weight), plan.dp(l.self_attn.in_proj_bias)
        a.out_proj_w, a.out_proj_b = plan.dp(l.self_attn.out_proj.weight), plan.dp(l.self_attn.out_proj.bias)
        a.norm2_g, a.norm2_b = plan.dp(l.norm2.weight), plan.dp(l.norm2.bias)
        a.lin1_w, a.lin1_b = plan.dp(l.linear1.weight), plan.dp(l.linear1.bias)
        a.lin2_w, a.lin2_b = plan.dp(l.linear2.weight), plan.dp(l.linear2.bias)
        if a.attn_in_packed:
            # all four matrices as the fragment stream of the token-stationary layer launch + the layer's vectors (layer_tok.hip)
            tp = torch.empty(lib().sf_layer_tok_packed_bytes(), dtype=torch.uint8, device=l.linear1.weight.device)
            check(lib().sf_pack_layer_tok_weights(C.byref(a), tp.data_ptr(), d, 8, ffn, torch.cuda.current_stream().cuda_stream))
            plan.keep.append(tp)
            a.tok_packed = tp.data_ptr()
    plan.keep.append(arr)
    return arr


# ---------------------------------------------------------------------------------------------
def rollouter_plan(r, packed=True):
    """r: SlotRollouter / SingleStepSlotRollouter container.  packed=False (training, train.py): plain torch-layout
    weights only -- the fragment-ordered copies the inference kernels read would have to be rebuilt after every
    optimizer step."""
    sig = _signature(r)
    attr = '_sf_plan' if packed else '_sf_train_plan'
    plan = getattr(r, attr, None)
    if plan is not None and plan.sig == sig:
        return plan
    plan = _Plan()
    s = sf_rollouter()
    single = hasattr(r, 'cond_len')
    W = r.cond_len if single else r.history_len
    N = r.num_slots
    s.num_slots, s.slot_size, s.d_model = N, r.in_proj.in_features, r.in_proj.out_features
    enc = r.transformer_encoder
    l0 = enc.layers[0]
    s.num_layers, s.num_heads = len(enc.layers), l0.self_attn.num_heads
    s.ffn_dim, s.norm_first = l0.linear1.out_features, int(l0.norm_first)
    s.window_len, s.single_step = W, int(single)
    s.in_proj_w, s.in_proj_b = plan.dp(r.in_proj.weight), plan.dp(r.in_proj.bias)
    s.out_proj_w, s.out_proj_b = plan.dp(r.out_proj.weight), plan.dp(r.out_proj.bias)
    # token PE: temporal PE repeated per slot (+ slots PE repeated per step)  slotformer.py:103-109
    pe = r.enc_t_pe.detach()[0].repeat_interleave(N, dim=0)
    if r.enc_slots_pe is not None:
        pe = pe + r.enc_slots_pe.detach()[0].repeat(W, 1)
    s.pe_tok = plan.dp(pe.contiguous())
    s.layers = C.cast(_tfm_layers(plan, enc, pack_ffn=packed), C.POINTER(sf_tfm_layer))
    if packed and s.d_model == 256 and s.slot_size == 128 and r.in_proj.weight.is_cuda:
        # fragment-ordered copies of in_proj / out_proj for the fused step-boundary kernel (layer_fused.hip)
        st = torch.cuda.current_stream().cuda_stream
        for name, lin in (('in_proj_packed', r.in_proj), ('out_proj_packed', r.out_proj)):
            n, k = lin.weight.shape
            buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=lin.weight.device)
            check(lib().sf_pack_linear_weights(plan.dp(lin.weight), buf.data_ptr(), n, k, st))
            plan.keep.append(buf)
            setattr(s, name, buf.data_ptr())
    plan.struct, plan.sig = s, sig
    setattr(r, attr, plan)
    return plan


def rollout_opts(opts):
    """dict / None -> ctypes sf_rollout_opts (None).  Keys: precision ('f32' | 'bf16x3' | 'bf16' | 'fp16' (probe) | 0..3), seam (bool),
    ffn_rows (32 | 64 | 128), attn_heads (2 | 8: heads per attention workgroup), attn_rows (0 | 128: q|k|v projection on row tiles of
    the batch + one core workgroup per video), ffn_tile (0 | 1 | 2: the FFN block as one workgroup per 64-row tile, finished rows; 2: fused with LN1 + q|k|v of the next layer), cus (CUs the calling stream's mask leaves it: seam launches only when their grid fits); per call and per thread, never process-wide."""
    if opts is None:
        return None
    if isinstance(opts, _lib.sf_rollout_opts):
        return opts
    unknown = set(opts) - {'precision', 'seam', 'ffn_rows', 'attn_heads', 'attn_rows', 'ffn_tile', 'cus', 'layer_tok'}
    if unknown:
        raise ValueError(f'slotformer_amd: unknown rollout options {sorted(unknown)}')
    prec = opts.get('precision', -1)
    prec = {'f32': 0, 'bf16x3': 1, 'bf16': 2, 'fp16': 3}.get(prec, prec)
    seam = opts.get('seam', None)
    return _lib.sf_rollout_opts(int(prec), -1 if seam is None else int(bool(seam)), int(opts.get('ffn_rows', 0)), int(opts.get('attn_heads', 0)),
                                int(opts.get('attn_rows', 0)), int(opts.get('ffn_tile', 0)), int(opts.get('cus', 0)),
                                {None: 0, True: 1, False: -1}[opts.get('layer_tok', None)])


def burn_in_of(r):
    """frames a rollout of `r` consumes: history_len, or 1 for the single-step rollouter (single_step_slotformer.py:49-63)"""
    return 1 if hasattr(r, 'cond_len') else r.history_len


def rollout(r, slots_all, n_in, pred_len, ws_slot=0, opts=None):
    """In-place autoregressive rollout.  slots_all [B, T_total, N, C] float32 contiguous on device;
    frames [0, n_in) hold the burn-in; frames [n_in, n_in+pred_len) are written.  n_in must be the rollouter's burn-in
    length (history_len; 1 for the single-step rollouter).  opts: see rollout_opts."""
    _require_inference(r, slots_all)
    ops._chk(slots_all)
    if n_in != burn_in_of(r):
        raise RuntimeError(f'slotformer_amd: burn-in of {n_in} frames, but this rollouter consumes {burn_in_of(r)} '
                           '(history_len; 1 for SingleStepSlotRollouter)')
    plan = rollouter_plan(r)
    B, T_total = slots_all.shape[:2]
    need = lib().sf_rollout_workspace_bytes(C.byref(plan.struct), B)
    ws = workspace(slots_all.device, need, ('roll', ws_slot))
    o = rollout_opts(opts)
    check(lib().sf_rollout_opts_f32(C.byref(plan.struct), slots_all.data_ptr(), B, T_total, pred_len, ws.data_ptr(),
                                    ws.numel(), torch.cuda.current_stream().cuda_stream, None if o is None else C.byref(o)))
    return slots_all


# ---------------------------------------------------------------------------------------------
def encoder_plan(m):
    """m: StoSAVi / STEVE container."""
    sig = _signature(m)
    plan = getattr(m, '_sf_plan', None)
    if plan is not None and plan.sig == sig:
        return plan
    plan = _Plan()
    s = sf_savi_encoder()
    if m.resolution[0] != m.resolution[1] or m.resolution[0] not in (64, 128):
        raise NotImplementedError(f'resolution {m.resolution}: the encoder needs 64x64 or 128x128 input '
                                  '(visual_resolution is fixed to 64x64, savi.py:226)')
    s.resolution = m.resolution[0]
    ch = list(m.enc_channels)
    n = len(ch) - 1
    if n > 8:
        raise NotImplementedError('at most 8 encoder convs')
    s.enc_layers, s.enc_ks = n, m.enc_ks
    for i, c in enumerate(ch):
        s.enc_channels[i] = c
    for i in range(n):
        conv = m.encoder[i][0]
        w = conv.weight.detach().float().contiguous()
        wp = w if i == 0 else ops.pack_conv_weight(w)
        s.conv_w[i] = plan.dp(wp)
        s.conv_b[i] = plan.dp(conv.bias)
        if i > 0 and w.is_cuda and tuple(w.shape) == (64, 64, 5, 5):
            # fragment-ordered split-bf16 copy: the conv on 4-row tiles with streamed weight fragments (conv_rows4.hip)
            frag = ops.pack_conv_frag(wp)   # (a byte buffer: kept alive by the plan, not converted by dp())
            plan.keep.append(frag)
            s.conv_w_frag[i] = frag.data_ptr()
    pe = m.encoder_pos_embedding
    s.pos_table = plan.dp(ops.pos_embed_table(pe.grid.detach().float(), pe.dense.weight.detach().float().contiguous(),
                                              pe.dense.bias.detach().float().contiguous()))
    eo = m.encoder_out_layer
    s.enc_ln_g, s.enc_ln_b = plan.dp(eo[0].weight), plan.dp(eo[0].bias)
    s.enc_fc1_w, s.enc_fc1_b = plan.dp(eo[1].weight), plan.dp(eo[1].bias)
    s.enc_fc2_w, s.enc_fc2_b = plan.dp(eo[3].weight), plan.dp(eo[3].bias)
    s.enc_out_channels = m.enc_out_channels
    s.num_slots, s.slot_size, s.slot_mlp_size = m.num_slots, m.slot_size, m.slot_mlp_size
    s.num_iterations = m.num_iterations
    sa = m.slot_attention
    s.sa_norm_in_g, s.sa_norm_in_b = plan.dp(sa.norm_inputs.weight), plan.dp(sa.norm_inputs.bias)
    s.sa_q_ln_g, s.sa_q_ln_b = plan.dp(sa.project_q[0].weight), plan.dp(sa.project_q[0].bias)
    s.sa_q_w = plan.dp(sa.project_q[1].weight)
    s.sa_q_w_t = plan.dp(sa.project_q[1].weight.detach().float().t().contiguous())   # [in, out] for the slot-update kernel
    s.sa_kv_w = plan.dp(torch.cat([sa.project_k.weight.detach(), sa.project_v.weight.detach()], 0).contiguous())
    tr = lambda w: w.detach().float().t().contiguous()  # noqa: E731  ([in, out] layout for the slot-update kernel)
    s.gru_w_ih, s.gru_w_hh = plan.dp(tr(sa.gru.weight_ih)), plan.dp(tr(sa.gru.weight_hh))
    s.gru_b_ih, s.gru_b_hh = plan.dp(sa.gru.bias_ih), plan.dp(sa.gru.bias_hh)
    s.mlp_ln_g, s.mlp_ln_b = plan.dp(sa.mlp[0].weight), plan.dp(sa.mlp[0].bias)
    s.mlp_w1, s.mlp_b1 = plan.dp(tr(sa.mlp[1].weight)), plan.dp(sa.mlp[1].bias)
    s.mlp_w2, s.mlp_b2 = plan.dp(tr(sa.mlp[3].weight)), plan.dp(sa.mlp[3].bias)
    if (m.slot_size, m.slot_mlp_size) in ((128, 256), (192, 384)) and sa.gru.weight_ih.is_cuda:
        # fragment-ordered split-bf16 copies for the matrix-core slot update (slot_update_mfma.hip; slot_update_wide.hip at 192)
        st = torch.cuda.current_stream().cuda_stream
        for name, w in (('sa_gru_ih_p', sa.gru.weight_ih), ('sa_gru_hh_p', sa.gru.weight_hh), ('sa_mlp_w1_p', sa.mlp[1].weight),
                        ('sa_mlp_w2_p', sa.mlp[3].weight), ('sa_q_w_p', sa.project_q[1].weight)):
            n, k = w.shape
            buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
            check(lib().sf_pack_linear_weights(plan.dp(w), buf.data_ptr(), n, k, st))
            plan.keep.append(buf)
            setattr(s, name, buf.data_ptr())
    if m.slot_size == m.enc_out_channels and m.slot_size in (128, 192) and sa.project_k.bias is None and sa.project_v.bias is None:
        # key / value projections folded into project_q and the GRU input matrix (include/slotformer_hip.h, sa_fold_*)
        wk, wv = sa.project_k.weight.detach().double(), sa.project_v.weight.detach().double()
        mq = (wk.t() @ sa.project_q[1].weight.detach().double()).float().contiguous()        # [C, D]
        gih = (sa.gru.weight_ih.detach().double() @ wv).float().contiguous()                 # [3D, C]
        s.sa_fold_q_w, s.sa_fold_q_w_t, s.sa_fold_gru_ih_t = plan.dp(mq), plan.dp(mq.t().contiguous()), plan.dp(gih.t().contiguous())
        if mq.is_cuda and m.slot_size == 192 and ch[-1] == 64:
            # the 192-wide per-pixel chain as one launch needs fragment-ordered copies of its two matrices (pixel_mlp.hip)
            st = torch.cuda.current_stream().cuda_stream
            for name, w in (('enc_fc1_p', eo[1].weight), ('enc_fc2_p', eo[3].weight)):
                n, k = w.shape
                buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
                check(lib().sf_pack_linear_weights(plan.dp(w), buf.data_ptr(), n, k, st))
                plan.keep.append(buf)
                setattr(s, name, buf.data_ptr())
        if mq.is_cuda and (m.slot_size, m.slot_mlp_size) in ((128, 256), (192, 384)):
            st = torch.cuda.current_stream().cuda_stream
            for name, w in (('sa_fold_q_w_p', mq), ('sa_fold_gru_ih_p', gih)):
                n, k = w.shape
                buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
                check(lib().sf_pack_linear_weights(plan.dp(w), buf.data_ptr(), n, k, st))
                plan.keep.append(buf)
                setattr(s, name, buf.data_ptr())
    s.init_latents = plan.dp(m.init_latents.detach()[0])
    s.sa_eps = float(sa.eps)
    kd = getattr(m, 'kernel_dist_layer', None)
    if kd is None:
        s.kd_mode = 0
    elif len(kd) == 1:
        s.kd_mode = 1
        s.kd_w0, s.kd_b0 = plan.dp(kd[0].weight), plan.dp(kd[0].bias)
        s.kd_w0_t = plan.dp(tr(kd[0].weight))   # [in, out] for the one-launch slot prologue
    else:
        s.kd_mode = 2
        s.kd_w0, s.kd_b0 = plan.dp(kd[0].weight), plan.dp(kd[0].bias)
        s.kd_ln_g, s.kd_ln_b = plan.dp(kd[1].weight), plan.dp(kd[1].bias)
        s.kd_w3, s.kd_b3 = plan.dp(kd[3].weight), plan.dp(kd[3].bias)
    pred = m.predictor
    rnn = hasattr(pred, 'rnn')
    base = pred.base_predictor if rnn else pred
    s.pred_rnn = int(rnn)
    s.pred_norm_first = int(base.norm_first)
    if hasattr(base, 'transformer_encoder'):
        s.pred_type = 1
        s.pred_num_layers, s.pred_num_heads, s.pred_ffn_dim = base.num_layers, base.num_heads, base.ffn_dim
        s.pred_layers = C.cast(_tfm_layers(plan, base.transformer_encoder), C.POINTER(sf_tfm_layer))
    else:
        s.pred_type = 0
        s.pred_ffn_dim = 2 * m.slot_size
        s.pm_ln_g, s.pm_ln_b = plan.dp(base.ln.weight), plan.dp(base.ln.bias)
        s.pm_w0, s.pm_b0 = plan.dp(base.mlp[0].weight), plan.dp(base.mlp[0].bias)
        s.pm_w2, s.pm_b2 = plan.dp(base.mlp[2].weight), plan.dp(base.mlp[2].bias)
        s.pm_w0_t, s.pm_w2_t = plan.dp(tr(base.mlp[0].weight)), plan.dp(tr(base.mlp[2].weight))
        if s.sa_mlp_w1_p and s.kd_mode == 1 and not rnn and base.mlp[0].weight.is_cuda and tuple(base.mlp[0].weight.shape) == (256, 128):
            # fragment-ordered copies for the next step's prologue at the tail of the matrix-core slot update (slot_update_mfma.hip, NEXT form)
            st = torch.cuda.current_stream().cuda_stream
            for name, w in (('pm_w0_p', base.mlp[0].weight), ('pm_w2_p', base.mlp[2].weight), ('kd_w0_p', kd[0].weight)):
                n, k = w.shape
                buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
                check(lib().sf_pack_linear_weights(plan.dp(w), buf.data_ptr(), n, k, st))
                plan.keep.append(buf)
                setattr(s, name, buf.data_ptr())
    if rnn:
        s.pred_hidden = pred.hidden_size
        s.lstm_w_ih, s.lstm_w_hh = plan.dp(pred.rnn.weight_ih_l0), plan.dp(pred.rnn.weight_hh_l0)
        s.lstm_b_ih, s.lstm_b_hh = plan.dp(pred.rnn.bias_ih_l0), plan.dp(pred.rnn.bias_hh_l0)
        s.proj_w, s.proj_b = plan.dp(pred.out_projector.weight), plan.dp(pred.out_projector.bias)
    if s.pred_type == 1 and base.norm_first and base.num_heads == 4 and m.init_latents.is_cuda:
        # fragment-ordered split-bf16 copies for the one-launch predictor step (pred_step.hip)
        mats = []
        for layer in base.transformer_encoder.layers:
            mats += [layer.self_attn.in_proj_weight, layer.self_attn.out_proj.weight, layer.linear1.weight, layer.linear2.weight]
        if rnn:
            mats += [pred.rnn.weight_ih_l0, pred.rnn.weight_hh_l0, pred.out_projector.weight]
        if all(w.shape[0] % 32 == 0 and w.shape[1] % 16 == 0 for w in mats):
            st = torch.cuda.current_stream().cuda_stream
            arr = (C.c_void_p * len(mats))()
            for i, w in enumerate(mats):
                n, k = w.shape
                buf = torch.empty(lib().sf_packed_linear_bytes(n, k), dtype=torch.uint8, device=w.device)
                check(lib().sf_pack_linear_weights(plan.dp(w), buf.data_ptr(), n, k, st))
                plan.keep.append(buf)
                arr[i] = buf.data_ptr()
            plan.keep.append(arr)
            s.pred_packed = C.cast(arr, C.POINTER(C.c_void_p))
    plan.struct, plan.sig = s, sig
    m._sf_plan = plan
    return plan


_FORK_STREAMS = {}


def _auto_side_stream(dev):
    """The second stream of direct (non-pipeline) encode calls: one per (device, host thread).  SF_ENCODE_FORK=0 turns the two-branch
    form off; it is also off while the current stream is being captured (the pipeline decides for its own graphs)."""
    if os.environ.get('SF_ENCODE_FORK', '1') == '0' or torch.cuda.is_current_stream_capturing():
        return None
    import threading
    key = (dev.index, threading.get_ident())
    st = _FORK_STREAMS.get(key)
    if st is None:
        st = _FORK_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


def savi_encode(m, img, prev_slots=None, noise=None, want_attn=False, ws_slot=0, feat_pre=None, side_stream='auto'):
    """Run StoSAVi.encode / STEVE.encode on device.  feat_pre [n_pre,B,4096,C]: CNN features of the first n_pre time
    steps computed ahead of time by `savi_cnn` (possibly on another stream).
    side_stream: a second torch stream -- the encode then runs as two branches (sf_savi_encode_fork_f32: the image features of the
    time steps on the current stream, the slot branches behind them on `side_stream`; same bits); the current stream continues
    behind both.  'auto' (default): a pooled stream per device and host thread (whole chip, nothing else running: 2.41 -> 2.17 ms per C2
    batch; 3.25 -> 2.81 on a 128-CU mask); None: one stream (what the batch pipeline uses: beside its CU-masked rollout streams a fifth
    busy hardware queue costs more than the fork saves, 497 -> 473 k frames/s, profiles/r04_probes.txt).

    Returns (post_slots [B,T,N,D], kernel_dist [B,T,N,2D] | None, attn [B,T,N,64*64] | None).
    The predictor's LSTM state lives on `m.predictor.hidden_state` exactly as in the reference.
    """
    _require_inference(m, img)
    img = img.float().contiguous()
    ops._chk(img, prev_slots, noise)
    plan = encoder_plan(m)
    s = plan.struct
    B, T = img.shape[:2]
    if tuple(img.shape[2:]) != (3, s.resolution, s.resolution):
        raise RuntimeError(f'img must be [B,T,3,{s.resolution},{s.resolution}], got {tuple(img.shape)}')
    N, D = s.num_slots, s.slot_size
    dev = img.device
    post = torch.empty(B, T, N, D, device=dev, dtype=torch.float32)
    kdist = torch.empty(B, T, N, 2 * D, device=dev, dtype=torch.float32) if s.kd_mode else None
    attn = torch.empty(B, T, N, 64 * 64, device=dev, dtype=torch.float32) if want_attn else None
    h = c = None
    valid = 0
    if s.pred_rnn:
        pred = m.predictor
        if prev_slots is None:
            pred.reset()  # StoSAVi._reset_rnn at the first frame (savi.py:474-475)
        st = pred.hidden_state
        if st is not None and st[0].shape[1] == B * N and st[0].device == dev:
            h, c, valid = st[0], st[1], 1
        else:
            h = torch.empty(1, B * N, s.pred_hidden, device=dev, dtype=torch.float32)
            c = torch.empty_like(h)
        pred.hidden_state = (h, c)
        pred.step += T if prev_slots is not None else T - 1
    if isinstance(side_stream, str):
        side_stream = _auto_side_stream(dev) if side_stream == 'auto' else None
    if side_stream is not None:
        need = lib().sf_savi_encode_fork_workspace_bytes(C.byref(plan.struct), B, T)
    elif feat_pre is None:
        # (room for the convolutions of all T steps as one launch per layer: csrc/engine.hip, batched form)
        need = lib().sf_savi_encode_batched_workspace_bytes(C.byref(plan.struct), B, T)
    else:
        need = lib().sf_savi_encode_workspace_bytes(C.byref(plan.struct), B)
    ws = workspace(dev, need, ('enc', ws_slot))
    P = ops._p
    n_pre = 0
    if feat_pre is not None:
        ops._chk(feat_pre)
        n_pre = feat_pre.shape[0]
        if feat_pre.shape[1] != B or feat_pre.shape[2] != 64 * 64:
            raise RuntimeError(f'feat_pre must be [n_pre,{B},4096,C], got {tuple(feat_pre.shape)}')
    check(lib().sf_savi_encode_fork_f32(C.byref(plan.struct), img.data_ptr(), P(feat_pre), n_pre, P(noise), P(prev_slots), P(h),
                                        P(c), valid, post.data_ptr(), P(kdist), P(attn), B, T, ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream().cuda_stream, None if side_stream is None else side_stream.cuda_stream))
    return post, kdist, attn


def savi_chain_ok(m, B, T):
    """True when the encode of this model can run in two halves (`savi_features` + `savi_slots_chain`: csrc/slot_chain.hip, the CLEVRER shape of
    StoSAVi's slot branch) at B videos x T frames."""
    plan = encoder_plan(m)
    return bool(lib().sf_savi_chain_ok(C.byref(plan.struct), int(B), int(T)))


def savi_planes_bytes(m, B, T):
    return int(lib().sf_savi_planes_bytes(C.byref(encoder_plan(m).struct), int(B), int(T)))


def savi_features(m, img, out, ws_slot=0):
    """First half of the encode (sf_savi_features_planes_f32): CNN + encoder_out_layer + SlotAttention.norm_inputs of img [B,T,3,R,R] -> `out`, a uint8
    tensor of savi_planes_bytes(m, B, T) bytes ([T][B][4096] rows of 512 B: bf16 hi | lo of the 128 channels), on the current stream.  No slots involved."""
    _require_inference(m, img)
    ops._chk(img)
    plan = encoder_plan(m)
    B, T = img.shape[:2]
    need = lib().sf_savi_features_workspace_bytes(C.byref(plan.struct), B, T)
    ws = workspace(img.device, need, ('encf', ws_slot))
    check(lib().sf_savi_features_planes_f32(C.byref(plan.struct), img.data_ptr(), B, T, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                            torch.cuda.current_stream().cuda_stream))
    return out


def savi_slots_chain(m, planes, nb, B, T, post, noise=None, prev_slots=None, kdist=None, attn=None, ws_slot=0):
    """Second half (sf_savi_slots_chain_f32): the slot branch of nb batches of B videos from `planes` ([nb][T][B][4096][512 B], uint8) as ONE launch of one
    workgroup per video.  post: a float32 tensor [nb * B, T', N, D] with T' >= T whose first T steps receive the slots (a rollout unit's buffer takes
    them in place); noise None or [nb * B, T, N, D]; prev_slots None or [nb * B, N, D]; kdist / attn: optional outputs [nb * B, T, N, 2 D] / [nb * B, T, N, 4096]."""
    plan = encoder_plan(m)
    s = plan.struct
    ops._chk(post, noise, prev_slots, kdist, attn)
    V = int(nb) * int(B)
    if post.shape[0] != V or post.shape[1] < T or post.stride(0) < T * s.num_slots * s.slot_size or post.stride(1) != s.num_slots * s.slot_size:
        raise RuntimeError(f'post must be [{V}, >= {T}, N, D] float32 with contiguous steps, got {tuple(post.shape)} strides {post.stride()}')
    need = lib().sf_savi_slots_chain_workspace_bytes(C.byref(plan.struct), V)
    ws = workspace(post.device, need, ('encs', ws_slot))
    P = ops._p
    check(lib().sf_savi_slots_chain_f32(C.byref(plan.struct), planes.data_ptr(), P(noise), P(prev_slots), post.data_ptr(), post.stride(0), P(kdist), P(attn),
                                        int(nb), int(B), int(T), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    return post


def savi_cnn(m, img, t0, t1, out=None, ws_slot=0):
    """CNN features (convs + soft position embedding) of time steps [t0, t1): [t1-t0, B, 4096, C_last] on the current
    stream; they do not depend on the slots, so they can be produced ahead of `savi_encode(..., feat_pre=...)`."""
    _require_inference(m, img)
    ops._chk(img)
    plan = encoder_plan(m)
    s = plan.struct
    B, T = img.shape[:2]
    cl = s.enc_channels[s.enc_layers]
    if out is None:
        out = torch.empty(t1 - t0, B, 64 * 64, cl, device=img.device, dtype=torch.float32)
    need = lib().sf_savi_cnn_workspace_bytes(C.byref(plan.struct), B)
    ws = workspace(img.device, need, ('cnn', ws_slot))
    check(lib().sf_savi_cnn_f32(C.byref(plan.struct), img.data_ptr(), B, T, t0, t1, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream))
    return out


# ---------------------------------------------------------------------------------------------
def _module_sig(*mods):
    return tuple((t.data_ptr(), t._version) for m in mods for t in list(m.parameters()) + list(m.buffers()))


def decoder_plan(m, inference=True):
    """m: StoSAVi or SlotFormer (both hold `decoder`, `decoder_pos_embedding`, dec_* attributes).  inference=False (the training nodes,
    whose parameters change every step): the torch-layout weights only -- no fragment-ordered copies, no fp64 fold of the first layer
    (a plan rebuilt per optimizer step must stay cheap: with the fold in it a StoSAVi training step took 121 instead of 21 ms)."""
    sig = _module_sig(m.decoder, m.decoder_pos_embedding)
    attr = '_sf_dec_plan' if inference else '_sf_dec_train_plan'
    plan = getattr(m, attr, None)
    if plan is not None and plan.sig == sig:
        return plan
    plan = _Plan()
    s = sf_savi_decoder()
    if m.resolution[0] != m.resolution[1] or m.dec_resolution[0] != m.dec_resolution[1]:
        raise NotImplementedError('square resolutions only')
    s.resolution, s.dec_res, s.dec_ks = m.resolution[0], m.dec_resolution[0], m.dec_ks
    s.num_slots, s.slot_size = m.num_slots, m.slot_size
    ch = list(m.dec_channels)
    n = len(ch) - 1
    s.dec_layers = n
    for i, c in enumerate(ch):
        s.dec_channels[i] = c
    for i in range(n):
        dc = m.decoder[i][0]
        s.dec_strides[i] = dc.stride[0]
        packed = ops.pack_deconv_weight(dc.weight.detach().float().contiguous())
        s.deconv_w[i] = plan.dp(packed)
        if dc.stride[0] == 1:   # [Cout, ks, ks, Cin] flipped over both kernel axes: the equivalent convolution kernel
            s.deconv_w_flipped[i] = plan.dp(packed.flip(1, 2).contiguous())
        s.deconv_b[i] = plan.dp(dc.bias)
        if inference and packed.is_cuda and tuple(dc.weight.shape) == (64, 64, 5, 5):
            st = torch.cuda.current_stream().cuda_stream
            if dc.stride[0] == 2:
                # consumption-ordered split-bf16 fragments: the parity-class kernel with streamed weights (deconv_s2.hip)
                frag = torch.empty(lib().sf_deconv_frag_bytes(64, 64, 5, 2), dtype=torch.uint8, device=packed.device)
                check(lib().sf_pack_deconv_frag_weights(packed.data_ptr(), frag.data_ptr(), 64, 64, 5, 2, st))
                plan.keep.append(frag)
                s.deconv_w_frag[i] = frag.data_ptr()
            elif dc.stride[0] == 1 and i == n - 1:
                # stride-1 last layer: the encoder's 4-row-tile convolution on the flipped kernel, 1x1 head in its epilogue (conv_rows4.hip)
                frag = ops.pack_conv_frag(packed.flip(1, 2).contiguous())
                plan.keep.append(frag)
                s.deconv_w_frag[i] = frag.data_ptr()
    dc0 = m.decoder[0][0]
    if inference and dc0.stride[0] == 2 and m.dec_ks == 5 and m.dec_resolution[0] >= 2 and dc0.bias is not None:
        # the first layer acts on slot + pos_table[p]: tap sums per border / parity class + the transposed convolution of the position
        # table (include/slotformer_hip.h, l0_weff / l0_posterm) -- a one-time fp64 fold on the host, exact algebra
        res0 = m.dec_resolution[0]
        w0 = dc0.weight.detach().double().cpu()                                  # [D, C1, 5, 5]
        taps = ([0, 2], [0, 2, 4], [2, 4], [1, 3], [3])                          # class -> taps whose input lies inside the map
        weff = torch.stack([torch.stack([w0[:, :, ky][:, :, :, kx].sum((2, 3)).t() for kx in taps]) for ky in taps])   # [5, 5, C1, D]
        pe0 = m.decoder_pos_embedding
        tab = ops.pos_embed_table(pe0.grid.detach().float(), pe0.dense.weight.detach().float().contiguous(),
                                  pe0.dense.bias.detach().float().contiguous()).double().cpu()                        # [res^2, D]
        pos_img = tab.view(res0, res0, -1).permute(2, 0, 1).unsqueeze(0)
        post = torch.nn.functional.conv_transpose2d(pos_img, w0, dc0.bias.detach().double().cpu(), stride=2, padding=2, output_padding=1)
        dev0 = dc0.weight.device
        s.l0_weff = plan.dp(weff.reshape(25 * w0.shape[1], w0.shape[0]).float().contiguous().to(dev0))
        s.l0_posterm = plan.dp(post[0].permute(1, 2, 0).reshape(4 * res0 * res0, w0.shape[1]).float().contiguous().to(dev0))
    head = m.decoder[n]
    s.out_w = plan.dp(head.weight.detach().float().reshape(head.out_channels, head.in_channels).contiguous())
    s.out_b = plan.dp(head.bias)
    pe = m.decoder_pos_embedding
    s.pos_table = plan.dp(ops.pos_embed_table(pe.grid.detach().float(), pe.dense.weight.detach().float().contiguous(),
                                              pe.dense.bias.detach().float().contiguous()))
    plan.struct, plan.sig = s, sig
    setattr(m, attr, plan)
    return plan


def savi_decode(m, slots, ws_slot=0, want=('recons', 'masks'), seg_dtype=torch.int64, fg_thre=0.5, out_recon=None, out_seg=None):
    """StoSAVi.decode on device: slots [F,N,D] -> (recon_combined [F,3,H,W], recons [F,N,3,H,W], masks [F,N,1,H,W]).
    `want`: which of the optional outputs to materialise -- 'recons', 'masks', 'seg' (postproc_mask of the decoded masks,
    vp_utils.py:20-41: [F,H,W] in `seg_dtype` int64 (the reference's) or uint8); with 'seg' the call returns a 4-tuple (.., seg);
    outputs not asked for are None and never written (a throughput caller that scores frames + segmentations skips 60 MB per frame).
    out_recon / out_seg: preallocated contiguous destinations ([F,3,H,W] float32 / [F,H,W] seg_dtype) instead of fresh tensors."""
    if torch.is_grad_enabled() and (slots.requires_grad or any(p.requires_grad for p in m.decoder.parameters())):
        from . import train
        return train.decode_with_grad(m, slots)   # one autograd node, data gradient only (row N1)
    if not slots.is_cuda:
        raise RuntimeError('slotformer_amd: inputs must live on a HIP device; there is no CPU fallback')
    slots = slots.detach().float().contiguous()
    plan = decoder_plan(m)
    F_, N, D = slots.shape
    if N > 16:
        raise NotImplementedError(f'slotformer_amd: savi_decode handles at most 16 slots per frame (got {N}): the recombination kernels keep a pixel\'s '
                                  'slot values in registers (csrc/elementwise.hip DC_NMAX)')
    H = plan.struct.resolution
    dev = slots.device
    recon = torch.empty(F_, 3, H, H, device=dev, dtype=torch.float32) if out_recon is None else out_recon
    if out_recon is not None and (tuple(recon.shape) != (F_, 3, H, H) or recon.dtype != torch.float32 or not recon.is_contiguous() or recon.device != dev):
        raise RuntimeError(f'out_recon must be a contiguous float32 [{F_},3,{H},{H}] tensor on {dev}')
    recons = torch.empty(F_, N, 3, H, H, device=dev, dtype=torch.float32) if 'recons' in want else None
    masks = torch.empty(F_, N, 1, H, H, device=dev, dtype=torch.float32) if 'masks' in want else None
    seg = None
    if 'seg' in want:
        if seg_dtype not in (torch.int64, torch.uint8):
            raise ValueError('slotformer_amd: seg_dtype must be torch.int64 or torch.uint8')
        seg = torch.empty(F_, H, H, device=dev, dtype=seg_dtype) if out_seg is None else out_seg
        if out_seg is not None and (tuple(seg.shape) != (F_, H, H) or seg.dtype != seg_dtype or not seg.is_contiguous() or seg.device != dev):
            raise RuntimeError(f'out_seg must be a contiguous {seg_dtype} [{F_},{H},{H}] tensor on {dev}')
    need = lib().sf_savi_decode_workspace_bytes(C.byref(plan.struct), F_)
    ws = workspace(dev, need, ('dec', ws_slot))
    P = ops._p
    check(lib().sf_savi_decode_seg_f32(C.byref(plan.struct), slots.data_ptr(), recon.data_ptr(), P(recons), P(masks),
                                       P(seg) if seg_dtype == torch.int64 else None, P(seg) if seg_dtype == torch.uint8 else None,
                                       float(fg_thre), F_, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    if 'seg' in want:
        return recon, recons, masks, seg
    return recon, recons, masks
