"""Plain PyTorch-CPU restatement of the reference hot path (TEST INFRASTRUCTURE).

Every function takes a *state dict* ``sd`` using the reference's own key names
 and returns tensors.  No nn.Module, no autograd, no GPU.
Citations are file:line under /root/reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import math

import torch
import torch.nn.functional as F

__all__ = [
    'build_grid', 'get_sin_pos_enc', 'layer_norm', 'gru_cell', 'lstm_cell',
    'mha_self', 'transformer_encoder_layer', 'transformer_encoder',
    'savi_encoder_out', 'slot_attention', 'slot_attention_updates', 'slot_update', 'predictor_step', 'kernel_dist',
    'sample_dist', 'kernel_kld', 'savi_encode', 'steve_encode', 'savi_forward_chunked',
    'rollouter_forward', 'rollouter_forward_train', 'single_step_rollouter_forward', 'slotformer_forward',
    'savi_decode', 'postproc_mask', 'rollout_video_slots', 'phyre_encode_rollout',
    'slot_mse_losses', 'dvae_logits', 'dvae_tokenize', 'dvae_detokenize', 'dvae_forward_train', 'steve_decoder_forward',
    'steve_decoder_generate', 'steve_forward_tokens', 'steve_slotformer_decode',
]


# ---------------------------------------------------------------------------
# closed forms
# ---------------------------------------------------------------------------
def build_grid(resolution, dtype=torch.float32):
    """[1,H,W,4] grid (y, x, 1-y, 1-x).  base_slots/models/utils.py:37-44."""
    ranges = [torch.linspace(0.0, 1.0, steps=r, dtype=dtype) for r in resolution]
    grid = torch.stack(torch.meshgrid(*ranges, indexing='ij'), dim=-1)
    grid = grid.reshape(resolution[0], resolution[1], -1).unsqueeze(0)
    return torch.cat([grid, 1.0 - grid], dim=-1)


def get_sin_pos_enc(seq_len, d_model):
    """[1,L,d]; newest frame has position 0.  video_prediction/models/slotformer.py:10-16."""
    inv_freq = 1. / (10000**(torch.arange(0.0, d_model, 2.0) / d_model))
    pos_seq = torch.arange(seq_len - 1, -1, -1).type_as(inv_freq)
    s = torch.outer(pos_seq, inv_freq)
    return torch.cat([s.sin(), s.cos()], dim=-1).unsqueeze(0)


# ---------------------------------------------------------------------------
# primitive ops (torch semantics the reference relies on)
# ---------------------------------------------------------------------------
def layer_norm(x, w, b, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1], ), w, b, eps)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """nn.GRUCell, gate order (r,z,n).  savi.py:48,95-99."""
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1. - z) * n + z * h


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One nn.LSTM step, gate order (i,f,g,o).  predictor.py:93,116-117."""
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, -1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def mha_self(x, in_w, in_b, out_w, out_b, nheads):
    """nn.MultiheadAttention self-attention, batch_first, no mask, eval."""
    B, L, d = x.shape
    hd = d // nheads
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.chunk(3, -1)
    q = q.view(B, L, nheads, hd).transpose(1, 2)
    k = k.view(B, L, nheads, hd).transpose(1, 2)
    v = v.view(B, L, nheads, hd).transpose(1, 2)
    att = torch.softmax((q * hd**-0.5) @ k.transpose(-1, -2), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, L, d)
    return F.linear(o, out_w, out_b)


def transformer_encoder_layer(x, sd, p, nheads, norm_first=True):
    """nn.TransformerEncoderLayer (relu, eval).  slotformer.py:72-80; predictor.py:33-38."""
    g = lambda k: sd[p + k]  # noqa: E731

    def sa(y):
        return mha_self(y, g('self_attn.in_proj_weight'), g('self_attn.in_proj_bias'),
                        g('self_attn.out_proj.weight'), g('self_attn.out_proj.bias'), nheads)

    def ff(y):
        return F.linear(F.relu(F.linear(y, g('linear1.weight'), g('linear1.bias'))),
                        g('linear2.weight'), g('linear2.bias'))

    if norm_first:
        x = x + sa(layer_norm(x, g('norm1.weight'), g('norm1.bias')))
        x = x + ff(layer_norm(x, g('norm2.weight'), g('norm2.bias')))
    else:
        x = layer_norm(x + sa(x), g('norm1.weight'), g('norm1.bias'))
        x = layer_norm(x + ff(x), g('norm2.weight'), g('norm2.bias'))
    return x


def transformer_encoder(x, sd, p, num_layers, nheads, norm_first=True):
    """nn.TransformerEncoder(norm=None)."""
    for i in range(num_layers):
        x = transformer_encoder_layer(x, sd, f'{p}layers.{i}.', nheads, norm_first)
    return x


# ---------------------------------------------------------------------------
# A1/A2: CNN encoder + soft position embedding + per-pixel MLP
# ---------------------------------------------------------------------------
def savi_encoder_out(img, sd, cfg):
    """img [F,3,H,W] -> [F, 64*64, enc_out].  savi.py:220-250, 367-377.

    CNN convention (nerv conv_norm_act, un-vendored): Conv2d(k, stride,
    padding=k//2, bias=True) -> Identity -> ReLU (none after last layer);
    state-dict keys encoder.{i}.0.{weight,bias}.  "parity unpinned" detail.
    """
    enc = cfg['enc_dict']
    ch = list(enc['enc_channels'])
    ks = enc['enc_ks']
    n = len(ch) - 1
    x = img
    for i in range(n):
        stride = 2 if (i == 0 and cfg['resolution'][0] == 128) else 1
        x = F.conv2d(x, sd[f'encoder.{i}.0.weight'], sd.get(f'encoder.{i}.0.bias'),
                     stride=stride, padding=ks // 2)
        if i != n - 1:
            x = F.relu(x)
    emb = F.linear(sd['encoder_pos_embedding.grid'], sd['encoder_pos_embedding.dense.weight'],
                   sd['encoder_pos_embedding.dense.bias'])  # [1,H,W,C]  utils.py:60-63
    x = x + emb.permute(0, 3, 1, 2)
    x = x.flatten(2, 3).permute(0, 2, 1).contiguous()
    x = layer_norm(x, sd['encoder_out_layer.0.weight'], sd['encoder_out_layer.0.bias'])
    x = F.relu(F.linear(x, sd['encoder_out_layer.1.weight'], sd['encoder_out_layer.1.bias']))
    x = F.linear(x, sd['encoder_out_layer.3.weight'], sd['encoder_out_layer.3.bias'])
    return x


# ---------------------------------------------------------------------------
# A3-A5: Slot Attention
# ---------------------------------------------------------------------------
def slot_attention(inputs, slots, sd, num_iterations, eps=1e-6, p='slot_attention.',
                   return_mask=False):
    """savi.py:56-102 (steve.py:19-73 with mask).  inputs [B,HW,C], slots [B,N,D]."""
    g = lambda k: sd[p + k]  # noqa: E731
    B, N, D = slots.shape
    x = layer_norm(inputs, g('norm_inputs.weight'), g('norm_inputs.bias'))
    k = F.linear(x, g('project_k.weight'))
    v = F.linear(x, g('project_v.weight'))
    scale = D**-0.5
    mask = None
    for it in range(num_iterations):
        prev = slots
        q = F.linear(layer_norm(slots, g('project_q.0.weight'), g('project_q.0.bias')),
                     g('project_q.1.weight'))
        logits = scale * torch.einsum('bnc,bmc->bnm', k, q)
        attn = torch.softmax(logits, dim=-1)
        if it == num_iterations - 1:
            mask = attn.permute(0, 2, 1).clone()
        attn = attn + eps
        attn = attn / attn.sum(dim=1, keepdim=True)
        upd = torch.einsum('bnm,bnc->bmc', attn, v)
        h = gru_cell(upd.reshape(B * N, D), prev.reshape(B * N, D), g('gru.weight_ih'),
                     g('gru.weight_hh'), g('gru.bias_ih'), g('gru.bias_hh')).view(B, N, D)
        m = layer_norm(h, g('mlp.0.weight'), g('mlp.0.bias'))
        m = F.linear(F.relu(F.linear(m, g('mlp.1.weight'), g('mlp.1.bias'))), g('mlp.3.weight'),
                     g('mlp.3.bias'))
        slots = h + m
    if return_mask:
        return slots, mask
    return slots


def slot_attention_updates(k, v, q, eps=1e-6):
    """The attention half of one iteration (savi.py:82-94): k, v [B,HW,D], q [B,N,D] -> updates [B,N,D].  Under
    autograd this is the gradient oracle of sf_slot_attn_iter_bwd_f32."""
    logits = q.shape[-1]**-0.5 * torch.einsum('bnc,bmc->bnm', k, q)
    attn = torch.softmax(logits, dim=-1) + eps
    attn = attn / attn.sum(dim=1, keepdim=True)
    return torch.einsum('bnm,bnc->bmc', attn, v)


def slot_update(updates, prev, sd, p='slot_attention.'):
    """The update half (savi.py:95-100): GRUCell then residual LayerNorm-MLP.  updates, prev [B,N,D] -> slots."""
    g = lambda k: sd[p + k]  # noqa: E731
    B, N, D = prev.shape
    h = gru_cell(updates.reshape(B * N, D), prev.reshape(B * N, D), g('gru.weight_ih'), g('gru.weight_hh'), g('gru.bias_ih'),
                 g('gru.bias_hh')).view(B, N, D)
    m = layer_norm(h, g('mlp.0.weight'), g('mlp.0.bias'))
    return h + F.linear(F.relu(F.linear(m, g('mlp.1.weight'), g('mlp.1.bias'))), g('mlp.3.weight'), g('mlp.3.bias'))


# ---------------------------------------------------------------------------
# A8/A9: kernel distribution and predictors
# ---------------------------------------------------------------------------
def kernel_dist(latents, sd, cfg):
    """savi.py:190-200."""
    if cfg['slot_dict'].get('kernel_mlp', True):
        x = F.linear(latents, sd['kernel_dist_layer.0.weight'], sd['kernel_dist_layer.0.bias'])
        x = F.relu(layer_norm(x, sd['kernel_dist_layer.1.weight'], sd['kernel_dist_layer.1.bias']))
        return F.linear(x, sd['kernel_dist_layer.3.weight'], sd['kernel_dist_layer.3.bias'])
    return F.linear(latents, sd['kernel_dist_layer.0.weight'], sd['kernel_dist_layer.0.bias'])


def _kld_method(cfg):
    return cfg['loss_dict']['kld_method'].split('-')[0]


def sample_dist(dist, cfg, noise):
    """savi.py:355-365.  ``noise`` replaces torch.randn_like (injected)."""
    D = cfg['slot_dict']['slot_size']
    mu = dist[..., :D]
    if _kld_method(cfg) == 'none':
        return mu
    return mu + noise * torch.exp(dist[..., D:] * 0.5)


def kernel_kld(dist, cfg):
    """StoSAVi._kld_loss, savi.py:338-353: KL(N(mu, var) || N(mu, prior_var)) summed over channels, mean over the rest."""
    if _kld_method(cfg) == 'none':
        return torch.zeros((), dtype=dist.dtype)
    import math
    D = cfg['slot_dict']['slot_size']
    parts = cfg['loss_dict']['kld_method'].split('-')
    prior = math.log(float(parts[1])) if len(parts) > 1 else 0.0
    log_var = dist[..., D:]
    kld = 0.5 * (prior - log_var) + torch.exp(log_var) / (2. * math.exp(prior)) - 0.5
    return kld.sum(-1).mean()


def predictor_step(x, sd, cfg, state):
    """predictor.py:20-135.  ``state`` = None | (h, c) carried LSTM state; returns (out, state)."""
    pd = cfg['pred_dict']
    rnn = pd['pred_rnn']
    bp = 'predictor.base_predictor.' if rnn else 'predictor.'
    if pd.get('pred_type', 'transformer') == 'mlp':
        ln = layer_norm(x, sd[bp + 'ln.weight'], sd[bp + 'ln.bias'])
        res = ln if pd['pred_norm_first'] else x  # predictor.py:65-73
        # channels = [D, 2D, D] -> mlp.0, mlp.2  (savi.py:300-303)
        out = F.linear(F.relu(F.linear(ln, sd[bp + 'mlp.0.weight'], sd[bp + 'mlp.0.bias'])),
                       sd[bp + 'mlp.2.weight'], sd[bp + 'mlp.2.bias']) + res
    else:
        out = transformer_encoder(x, sd, bp + 'transformer_encoder.', pd['pred_num_layers'],
                                  pd['pred_num_heads'], pd['pred_norm_first'])
    if not rnn:
        return out, state
    shp = out.shape
    flat = out.reshape(-1, shp[-1])
    hid = sd['predictor.rnn.weight_hh_l0'].shape[1]
    if state is None:
        state = (flat.new_zeros(flat.shape[0], hid), flat.new_zeros(flat.shape[0], hid))
    h, c = lstm_cell(flat, state[0], state[1], sd['predictor.rnn.weight_ih_l0'],
                     sd['predictor.rnn.weight_hh_l0'], sd['predictor.rnn.bias_ih_l0'],
                     sd['predictor.rnn.bias_hh_l0'])
    out = F.linear(h, sd['predictor.out_projector.weight'],
                   sd['predictor.out_projector.bias']).view(shp)
    return out, (h, c)


# ---------------------------------------------------------------------------
# A6/A7: encode loops
# ---------------------------------------------------------------------------
def savi_encode(img, sd, cfg, prev_slots=None, state=None, noise=None):
    """StoSAVi.encode, savi.py:379-416.  img [B,T,3,H,W]; noise [B,T,N,D] or None.

    Returns dict(post_slots, kernel_dist, encoder_out, state).
    """
    B, T = img.shape[:2]
    N, D = cfg['slot_dict']['num_slots'], cfg['slot_dict']['slot_size']
    enc = savi_encoder_out(img.flatten(0, 1), sd, cfg).unflatten(0, (B, T))
    init = sd['init_latents'].repeat(B, 1, 1)
    dists, slots_all = [], []
    for t in range(T):
        if prev_slots is None:
            latents = init
        else:
            latents, state = predictor_step(prev_slots, sd, cfg, state)
        kd = kernel_dist(latents, sd, cfg)
        kernels = sample_dist(kd, cfg, None if noise is None else noise[:, t])
        post = slot_attention(enc[:, t], kernels, sd, cfg['slot_dict']['num_iterations'])
        dists.append(kd)
        slots_all.append(post)
        prev_slots = post
    return dict(post_slots=torch.stack(slots_all, 1), kernel_dist=torch.stack(dists, 1),
                encoder_out=enc, state=state)


def steve_encode(img, sd, cfg, prev_slots=None, state=None, training=False):
    """STEVE.encode, steve.py:198-240.  Returns dict(slots, masks, state)."""
    B, T = img.shape[:2]
    enc = savi_encoder_out(img.flatten(0, 1), sd, cfg).unflatten(0, (B, T))
    N = cfg['slot_dict']['num_slots']
    slots_all, masks_all = [], []
    for t in range(T):
        if prev_slots is None:
            latents = sd['init_latents'].repeat(B, 1, 1)
        else:
            latents, state = predictor_step(prev_slots, sd, cfg, state)
        s, m = slot_attention(enc[:, t], latents, sd, cfg['slot_dict']['num_iterations'],
                              return_mask=True)
        slots_all.append(s)
        masks_all.append(m.unflatten(-1, (64, 64)))
        prev_slots = s
    slots = torch.stack(slots_all, 1)
    masks = torch.stack(masks_all, 1).contiguous()
    res = tuple(cfg['resolution'])
    if (not training) and res != (64, 64):
        masks = F.interpolate(masks.flatten(0, 2).unsqueeze(1), res, mode='bilinear',
                              align_corners=False).squeeze(1).unflatten(0, (B, T, N))
    return dict(slots=slots, masks=masks, state=state)


def savi_forward_chunked(img, sd, cfg, clip_len, noise=None):
    """StoSAVi.forward long-video path, savi.py:431-463: chunk over T carrying
    prev_slots; predictor state is reset only when prev_slots is None (:474-475)."""
    T = img.shape[1]
    prev, state = None, None
    posts, dists = [], []
    for c0 in range(0, T, clip_len):
        out = savi_encode(img[:, c0:c0 + clip_len], sd, cfg, prev_slots=prev,
                          state=state if prev is not None else None,
                          noise=None if noise is None else noise[:, c0:c0 + clip_len])
        posts.append(out['post_slots'])
        dists.append(out['kernel_dist'])
        prev, state = out['post_slots'][:, -1], out['state']
    return dict(post_slots=torch.cat(posts, 1), kernel_dist=torch.cat(dists, 1))


# ---------------------------------------------------------------------------
# A10-A14: rollout
# ---------------------------------------------------------------------------
def rollouter_forward(x, pred_len, sd, rcfg, p='rollouter.'):
    """SlotRollouter.forward, slotformer.py:85-126.  x [B,hist,N,C] -> [B,pred_len,N,C]."""
    B, hist, N, C = x.shape
    assert hist == rcfg['history_len'], 'wrong burn-in steps'
    in_x = x.flatten(1, 2)
    pe = sd[p + 'enc_t_pe'].unsqueeze(2).repeat(B, 1, N, 1).flatten(1, 2)
    if rcfg.get('slots_pe'):
        pe = pe + sd[p + 'enc_slots_pe'].unsqueeze(1).repeat(B, hist, 1, 1).flatten(1, 2)
    out = []
    for _ in range(pred_len):
        h = F.linear(in_x, sd[p + 'in_proj.weight'], sd[p + 'in_proj.bias']) + pe
        h = transformer_encoder(h, sd, p + 'transformer_encoder.', rcfg['num_layers'],
                                rcfg['num_heads'], rcfg['norm_first'])
        pred = F.linear(h[:, -N:], sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])
        out.append(pred)
        in_x = torch.cat([in_x[:, N:], pred], dim=1)
    return torch.stack(out, 1)


def single_step_rollouter_forward(x, pred_len, sd, rcfg, p='rollouter.'):
    """SingleStepSlotRollouter.forward, single_step_slotformer.py:49-90."""
    B, hist, N, C = x.shape
    assert hist == rcfg['history_len'] == 1
    cond = rcfg['cond_len']
    in_x = x.flatten(1, 2)
    pe = sd[p + 'enc_t_pe'].unsqueeze(2).repeat(B, 1, N, 1).flatten(1, 2)
    if rcfg.get('slots_pe'):
        pe = pe + sd[p + 'enc_slots_pe'].unsqueeze(1).repeat(B, cond, 1, 1).flatten(1, 2)
    out = []
    for _ in range(pred_len):
        h = F.linear(in_x[:, -cond * N:], sd[p + 'in_proj.weight'], sd[p + 'in_proj.bias'])
        h = h + pe[:, -h.shape[1]:]
        h = transformer_encoder(h, sd, p + 'transformer_encoder.', rcfg['num_layers'],
                                rcfg['num_heads'], rcfg['norm_first'])
        pred = F.linear(h[:, -N:], sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])
        out.append(pred)
        in_x = torch.cat([in_x, pred], dim=1)
    return torch.stack(out, 1)


def rollouter_forward_train(x, pred_len, sd, rcfg, drop, p='rollouter.'):
    """SlotRollouter.forward in train() mode (slotformer.py:85-126): the four dropout sites of torch's pre-LN
    nn.TransformerEncoderLayer made explicit -- 0: softmax weights inside nn.MultiheadAttention, 1: attention block output
    (dropout1), 2: FFN hidden after ReLU (dropout), 3: FFN output (dropout2).  `drop(step, layer, site, t)` returns the
    dropped tensor (t * mask / keep); torch ops throughout, so autograd of this function is the gradient oracle."""
    B, hist, N, C = x.shape
    assert hist == rcfg['history_len'] and rcfg['norm_first']
    H, nl = rcfg['num_heads'], rcfg['num_layers']
    in_x = x.flatten(1, 2)
    pe = sd[p + 'enc_t_pe'].unsqueeze(2).repeat(B, 1, N, 1).flatten(1, 2)
    if rcfg.get('slots_pe'):   # slotformer.py:106-109
        pe = pe + sd[p + 'enc_slots_pe'].unsqueeze(1).repeat(B, hist, 1, 1).flatten(1, 2)
    out = []
    for s in range(pred_len):
        h = F.linear(in_x, sd[p + 'in_proj.weight'], sd[p + 'in_proj.bias']) + pe
        for l in range(nl):
            g = lambda k: sd[f'{p}transformer_encoder.layers.{l}.{k}']  # noqa: E731
            d = h.shape[-1]
            hd = d // H
            y = layer_norm(h, g('norm1.weight'), g('norm1.bias'))
            q, k, v = F.linear(y, g('self_attn.in_proj_weight'), g('self_attn.in_proj_bias')).chunk(3, -1)
            q, k, v = (t.view(B, -1, H, hd).transpose(1, 2) for t in (q, k, v))
            att = drop(s, l, 0, torch.softmax((q * hd**-0.5) @ k.transpose(-1, -2), dim=-1))
            o = (att @ v).transpose(1, 2).reshape(B, -1, d)
            h = h + drop(s, l, 1, F.linear(o, g('self_attn.out_proj.weight'), g('self_attn.out_proj.bias')))
            y = layer_norm(h, g('norm2.weight'), g('norm2.bias'))
            y = drop(s, l, 2, F.relu(F.linear(y, g('linear1.weight'), g('linear1.bias'))))
            h = h + drop(s, l, 3, F.linear(y, g('linear2.weight'), g('linear2.bias')))
        pred = F.linear(h[:, -N:], sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])
        out.append(pred)
        in_x = torch.cat([in_x[:, N:], pred], dim=1)
    return torch.stack(out, 1)


def slotformer_forward(slots, sd, cfg, rollout_len, single_step=False):
    """SlotFormer.forward (no image decode), slotformer.py:263-282."""
    rcfg = cfg['rollout_dict']
    hist = rcfg['history_len']
    assert rollout_len + hist == slots.shape[1]
    fn = single_step_rollouter_forward if single_step else rollouter_forward
    pred = fn(slots[:, :hist][:, -hist:], rollout_len, sd, rcfg)
    return dict(pred_slots=pred, gt_slots=slots[:, hist:])


def slot_mse_losses(pred, gt, training, loss_decay_factor=1., vid_len=None, history_len=0):
    """SlotFormer.calc_train_loss (slot part), slotformer.py:284-318."""
    out = {}
    loss = F.mse_loss(pred, gt, reduction='none')
    if not training:
        for s in range(min(6, gt.shape[1])):
            out[f'slot_recon_loss_{s + 1}'] = loss[:, s].mean()
    if loss_decay_factor < 1.:
        w = (loss_decay_factor**torch.arange(gt.shape[1])).type_as(loss)
        w = w / w.sum() * gt.shape[1]
        loss = loss * w[None, :, None, None]
    if vid_len is not None and (vid_len < history_len + gt.shape[1]).any():
        valid = (torch.arange(gt.shape[1]) + history_len)[None] < vid_len[:, None]
        loss = loss.flatten(0, 1)[valid.flatten(0, 1)]
    out['slot_recon_loss'] = loss.mean()
    return out


# ---------------------------------------------------------------------------
# "next" rows: decoder (N2), masks (M1), harness index arithmetic (H2/H3)
# ---------------------------------------------------------------------------
def savi_decode(slots, sd, cfg):
    """StoSAVi.decode, savi.py:504-525.  slots [F,N,D] -> (recon [F,3,H,W], recons, masks).

    Deconv convention (nerv deconv_norm_act, un-vendored): ConvTranspose2d(k,
    stride, padding=k//2, output_padding=stride-1, bias=True) -> ReLU.
    """
    dec = cfg['dec_dict']
    ch = list(dec['dec_channels'])
    ks = dec['dec_ks']
    res = cfg['resolution'][0]
    Fr, N, D = slots.shape
    r = dec['dec_resolution']
    x = slots.reshape(Fr * N, D, 1, 1).repeat(1, 1, r[0], r[1])
    emb = F.linear(sd['decoder_pos_embedding.grid'], sd['decoder_pos_embedding.dense.weight'],
                   sd['decoder_pos_embedding.dense.bias'])
    x = x + emb.permute(0, 3, 1, 2)
    out_size, stride = r[0], 2
    for i in range(len(ch) - 1):
        if out_size == res:
            stride = 1
        x = F.relu(F.conv_transpose2d(x, sd[f'decoder.{i}.0.weight'], sd.get(f'decoder.{i}.0.bias'),
                                      stride=stride, padding=ks // 2, output_padding=stride - 1))
        out_size = (out_size - 1) * stride - 2 * (ks // 2) + ks + stride - 1
    j = len(ch) - 1
    x = F.conv2d(x, sd[f'decoder.{j}.weight'], sd[f'decoder.{j}.bias'])
    x = x.view(Fr, N, 4, res, res)
    recons, masks = x[:, :, :3], torch.softmax(x[:, :, 3:], dim=1)
    return (recons * masks).sum(1), recons, masks


def postproc_mask(batch_masks, fg_thre=0.5):
    """vp_utils.py:20-41.  [B,T,N,1,H,W] -> int64 [B,T,H,W]."""
    m = batch_masks.clone()
    B, T, N, _, H, W = m.shape
    m = m.reshape(B * T, N, H * W)
    bg_idx = m.max(-1)[0].argmin(-1)
    bg_mask = m.max(1)[0] < fg_thre
    idx = torch.zeros(B * T, N, dtype=torch.bool)
    idx[torch.arange(B * T), bg_idx] = True
    m[idx.unsqueeze(-1) * bg_mask.unsqueeze(1)] = 1.
    return m.argmax(1).reshape(B, T, H, W)


def rollout_video_slots(ori_slots, sd, cfg, frame_offset, obs_frames=128, target_len=160):
    """rollout_clevrer_slots.py:20-65 index arithmetic.  ori_slots [B,obs,N,C] -> [B,target,N,C]."""
    hist = cfg['rollout_dict']['history_len']
    B, _, N, C = ori_slots.shape
    pad = torch.zeros(B, target_len - obs_frames, N, C).type_as(ori_slots)
    full = torch.cat([ori_slots, pad], 1)
    preds = []
    for off in range(frame_offset):
        start = obs_frames - hist * frame_offset + off
        in_slots = full[:, start::frame_offset]
        preds.append(slotformer_forward(in_slots, sd, cfg, in_slots.shape[1] - hist)['pred_slots'])
    pred = torch.stack([preds[i % frame_offset][:, i // frame_offset]
                        for i in range(target_len - obs_frames)], 1)
    return torch.cat([full[:, :obs_frames], pred], 1)


def phyre_encode_rollout(img0, savi_sd, savi_cfg, sf_sd, sf_cfg, vid_len, noise=None):
    """test_phyre_planning.py:159-174: SAVi on frame 0 -> zero-pad -> SingleStepSlotFormer."""
    slot0 = savi_encode(img0, savi_sd, savi_cfg, noise=noise)['post_slots']  # [B,1,N,C]
    B, _, N, C = slot0.shape
    slots = torch.zeros(B, vid_len, N, C).type_as(slot0)
    slots[:, :1] = slot0
    return slotformer_forward(slots, sf_sd, sf_cfg, vid_len - 1, single_step=True)


# ---------------------------------------------------------------------------
# STEVE image side (row N2, second half): dVAE and the slot-conditioned Transformer decoder
# ---------------------------------------------------------------------------
def _conv_block(x, sd, p, stride=1, padding=0):
    """Conv2dBlock: bias-free conv -> GroupNorm(1 group, affine) -> ReLU.  steve_utils.py:100-126."""
    x = F.conv2d(x, sd[p + 'm.weight'], None, stride, padding)
    return F.relu(F.group_norm(x, 1, sd[p + 'weight'], sd[p + 'bias']))


def dvae_logits(img, sd, p=''):
    """img [F,3,H,W] -> vocabulary logits [F,V,H/4,W/4].  dVAE.py:24-34 (encoder stack)."""
    x = _conv_block(img, sd, p + 'encoder.0.', stride=4)
    for i in range(1, 7):
        x = _conv_block(x, sd, p + f'encoder.{i}.')
    return F.conv2d(x, sd[p + 'encoder.7.weight'], sd[p + 'encoder.7.bias'])


def dvae_tokenize(img, sd, p='', one_hot=True):
    """dVAE.tokenize (dVAE.py:52-77): argmax ids [F,h,w], or the one-hot map [F,V,h,w] (steve_utils.py:10-13)."""
    logits = dvae_logits(img, sd, p)
    idx = logits.argmax(dim=1)
    if not one_hot:
        return idx
    return torch.zeros_like(logits).scatter_(1, idx.unsqueeze(1), 1.)


def dvae_detokenize(z, sd, p=''):
    """z [F,V,h,w] (probabilities / one-hot) -> image [F,3,4h,4w].  dVAE.py:36-50,79-100 (decoder stack)."""
    x = _conv_block(z, sd, p + 'decoder.0.')
    x = _conv_block(x, sd, p + 'decoder.1.', padding=1)
    x = _conv_block(x, sd, p + 'decoder.2.')
    x = _conv_block(x, sd, p + 'decoder.3.')
    x = F.pixel_shuffle(_conv_block(x, sd, p + 'decoder.4.'), 2)
    x = _conv_block(x, sd, p + 'decoder.6.', padding=1)
    x = _conv_block(x, sd, p + 'decoder.7.')
    x = _conv_block(x, sd, p + 'decoder.8.')
    x = F.pixel_shuffle(_conv_block(x, sd, p + 'decoder.9.'), 2)
    return F.conv2d(x, sd[p + 'decoder.11.weight'], sd[p + 'decoder.11.bias'])


def dvae_forward_train(img, gumbel, sd, tau=1., hard=False, p=''):
    """dVAE.forward outside testing (dVAE.py:113-139) with the Gumbel noise of steve_utils.py:26-41 given: logits ->
    log_softmax -> softmax((z_logits + gumbel) / tau) (straight-through one-hot when `hard`) -> decoder; the loss is
    F.mse_loss(recon, img) (dVAE.py:141-146).  img [F,3,H,W], gumbel [F,V,h,w]."""
    z_logits = F.log_softmax(dvae_logits(img, sd, p), dim=1)
    y_soft = F.softmax((z_logits + gumbel) / tau, 1)
    if hard:
        y_hard = torch.zeros_like(y_soft).scatter_(1, y_soft.argmax(1, keepdim=True), 1.)
        z = y_hard - y_soft.detach() + y_soft
    else:
        z = y_soft
    recon = dvae_detokenize(z, sd, p)
    return {'recon': recon, 'z_logits': z_logits, 'recon_loss': F.mse_loss(recon, img)}


def _slate_mha(q, k, v, sd, p, nheads, mask=None):
    """MultiHeadAttention of steve_transformer.py:12-55: bias-free projections, q scaled by hd^-0.5, optional
    boolean mask (True = blocked)."""
    B, T, d = q.shape
    S = k.shape[1]
    hd = d // nheads
    q = (q @ sd[p + 'proj_q.weight'].t()).view(B, T, nheads, hd).transpose(1, 2) * hd**-0.5
    k = (k @ sd[p + 'proj_k.weight'].t()).view(B, S, nheads, hd).transpose(1, 2)
    v = (v @ sd[p + 'proj_v.weight'].t()).view(B, S, nheads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2)
    if mask is not None:
        att = att.masked_fill(mask, float('-inf'))
    out = (att.softmax(-1) @ v).transpose(1, 2).reshape(B, T, d)
    return out @ sd[p + 'proj_o.weight'].t()


def steve_decoder_forward(slots, idx, sd, nheads, num_layers, p='trans_decoder.'):
    """STEVETransformerDecoder.forward (steve_transformer.py:275-303): slots [B,N,d], idx [B,t] int64 (without the last
    target token) -> logits [B,1+t,V].  Blocks: steve_transformer.py:146-199 (first block normalises its input in
    place, :186-190)."""
    B, T = idx.shape
    V = sd[p + 'head.weight'].shape[0]
    mem = slots @ sd[p + 'in_proj.weight'].t() + sd[p + 'in_proj.bias']
    idx = torch.cat([torch.full((B, 1), V, dtype=idx.dtype), idx], 1)
    x = sd[p + 'tok_emb.weight'][idx] + sd[p + 'pos_emb.pe'][:, :T + 1]
    L = T + 1
    causal = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1)
    for i in range(num_layers):
        q = p + f'tf_dec.blocks.{i}.'
        ln = lambda t, name: layer_norm(t, sd[q + name + '.weight'], sd[q + name + '.bias'])  # noqa: E731
        if i == 0:
            x = ln(x, 'self_attn_layer_norm')
            x = x + _slate_mha(x, x, x, sd, q + 'self_attn.', nheads, causal)
        else:
            y = ln(x, 'self_attn_layer_norm')
            x = x + _slate_mha(y, y, y, sd, q + 'self_attn.', nheads, causal)
        y = ln(x, 'encoder_decoder_attn_layer_norm')
        x = x + _slate_mha(y, mem, mem, sd, q + 'encoder_decoder_attn.', nheads)
        y = ln(x, 'ffn_layer_norm')
        y = F.relu(y @ sd[q + 'ffn.0.weight'].t() + sd[q + 'ffn.0.bias'])
        x = x + y @ sd[q + 'ffn.2.weight'].t() + sd[q + 'ffn.2.bias']
    x = layer_norm(x, sd[p + 'tf_dec.layer_norm.weight'], sd[p + 'tf_dec.layer_norm.bias'])
    return x @ sd[p + 'head.weight'].t()


def steve_decoder_generate(slots, steps, sd, nheads, num_layers, p='trans_decoder.'):
    """STEVETransformerDecoder.generate with sample=False (steve_transformer.py:305-333): greedy tokens [B,steps] and
    the per-step logits [B,steps,V]."""
    B = slots.shape[0]
    idx = torch.zeros(B, 0, dtype=torch.long)
    logits_all = []
    for _ in range(steps):
        lg = steve_decoder_forward(slots, idx, sd, nheads, num_layers, p)[:, -1]
        logits_all.append(lg)
        idx = torch.cat([idx, lg.argmax(-1, keepdim=True)], 1)
    return idx, torch.stack(logits_all, 1)


def steve_forward_tokens(img, slots, sd, cfg, gumbel=None):
    """The token-prediction half of STEVE._forward (steve.py:306-322): dVAE ids of the frames as targets, teacher-forced
    decoder logits, and the cross-entropy of steve.py:341-344.  img [B,T,3,H,W], slots [B,T,N,D].  With `gumbel`
    [B*T,V,h,w] also the optional image term (use_img_recon_loss, steve.py:327-335, 345-349): relaxed sample of the predicted
    token map at tau 0.1, decoded by the frozen dVAE, MSE against the frames."""
    tgt = dvae_tokenize(img.flatten(0, 1), sd, 'dvae.', one_hot=False).flatten(1, 2)
    dd = cfg['dec_dict']
    logits = steve_decoder_forward(slots.flatten(0, 1), tgt[:, :-1], sd, dd['dec_num_heads'], dd['dec_num_layers'])
    loss = F.cross_entropy(logits.flatten(0, 1), tgt.flatten(0, 1))
    out = {'pred_token_id': logits, 'target_token_id': tgt, 'token_recon_loss': loss}
    if gumbel is not None:
        h, w = gumbel.shape[-2:]
        z_logits = F.log_softmax(logits.transpose(2, 1).unflatten(-1, (h, w)), dim=1)
        z = F.softmax((z_logits + gumbel) / 0.1, 1)
        out['recon_img'] = dvae_detokenize(z, sd, 'dvae.')
        out['img_recon_loss'] = F.mse_loss(out['recon_img'], img.flatten(0, 1))
    return out


def steve_slotformer_decode(slots, sd, cfg, gumbel):
    """STEVESlotFormer.decode (steve_slotformer.py:86-103): greedy token generation with the frozen decoder, the
    Gumbel-softmax relaxation at tau = 0.1 with the noise `gumbel` [B,V,h,w] injected, dVAE detokenisation of the soft
    and of the one-hot token maps."""
    dd = cfg['dec_dict']
    h = cfg['resolution'][0] // cfg['dvae_dict']['down_factor']
    w = cfg['resolution'][1] // cfg['dvae_dict']['down_factor']
    _, logits = steve_decoder_generate(slots, h * w, sd, dd['dec_num_heads'], dd['dec_num_layers'], p='decoder.')
    logits = logits.transpose(2, 1).unflatten(-1, (h, w)).contiguous()
    z = F.softmax((F.log_softmax(logits, dim=1) + gumbel) / 0.1, dim=1)
    soft = dvae_detokenize(z, sd, 'dvae.')
    hard = dvae_detokenize(torch.zeros_like(logits).scatter_(1, logits.argmax(1, keepdim=True), 1.), sd, 'dvae.')
    return soft, hard
