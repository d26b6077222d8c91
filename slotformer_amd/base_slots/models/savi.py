"""SAVi / StoSAVi slot-extraction model on the MI355X engine.

Mirrors the public surface of the reference's slotformer/base_slots/models/savi.py
(constructor arguments, attribute names, output dicts, state-dict keys) so checkpoints and the
extraction scripts work unchanged.  The nn.Modules below only *hold parameters*; the forward
arithmetic runs in libslotformer_hip (sf_savi_encode_f32).
"""
import math

import torch
from torch import nn

from ...nerv_compat import BaseModel
from ... import engine, ops
from ...host import containers, losses
from .utils import assert_shape, SoftPositionEmbed
from .predictor import ResidualMLPPredictor, TransformerPredictor, RNNPredictorWrapper


class SlotAttention(nn.Module):
    """Slot Attention parameters (reference savi.py:16-110).

    forward(inputs [B,HW,C], slots [B,N,D]) runs the iteration kernels:
    q-projection (fused LN GEMM) -> sf_slot_attn_iter_f32 -> sf_slot_update_f32.
    """

    def __init__(self, in_features, num_iterations, num_slots, slot_size, mlp_hidden_size, eps=1e-6):
        super().__init__()
        self.in_features = in_features
        self.num_iterations = num_iterations
        self.num_slots = num_slots
        self.slot_size = slot_size
        self.mlp_hidden_size = mlp_hidden_size
        self.eps = eps
        self.attn_scale = self.slot_size**-0.5

        self.norm_inputs = nn.LayerNorm(self.in_features)
        self.project_q = nn.Sequential(
            nn.LayerNorm(self.slot_size),
            nn.Linear(self.slot_size, self.slot_size, bias=False),
        )
        self.project_k = nn.Linear(in_features, self.slot_size, bias=False)
        self.project_v = nn.Linear(in_features, self.slot_size, bias=False)
        self.gru = nn.GRUCell(self.slot_size, self.slot_size)
        self.mlp = containers.dense_stack([self.slot_size, self.mlp_hidden_size, self.slot_size], norm_first=True)

    def _run(self, inputs, slots, want_mask):
        engine._require_inference(self, inputs, slots)
        assert len(slots.shape) == 3
        inputs = inputs.float().contiguous()
        slots = slots.float().contiguous()
        d = lambda t: t.detach()  # noqa: E731
        kv_w = torch.cat([d(self.project_k.weight), d(self.project_v.weight)], 0).contiguous()
        kv = ops.linear(inputs, kv_w, ln=(d(self.norm_inputs.weight), d(self.norm_inputs.bias)))
        k, v = kv[..., :self.slot_size].contiguous(), kv[..., self.slot_size:].contiguous()
        mask = None
        for it in range(self.num_iterations):
            q = ops.linear(slots, d(self.project_q[1].weight),
                           ln=(d(self.project_q[0].weight), d(self.project_q[0].bias)))
            last = it == self.num_iterations - 1
            pn, pd, attn = ops.slot_attn_iter(k, v, q, eps=self.eps, want_attn=want_mask and last)
            if attn is not None:
                mask = attn
            slots = ops.slot_update(
                pn, pd, slots, (d(self.gru.weight_ih), d(self.gru.weight_hh), d(self.gru.bias_ih), d(self.gru.bias_hh)),
                d(self.mlp[0].weight), d(self.mlp[0].bias), d(self.mlp[1].weight), d(self.mlp[1].bias),
                d(self.mlp[3].weight), d(self.mlp[3].bias))
        return slots, mask

    def forward(self, inputs, slots):
        if torch.is_grad_enabled() and (inputs.requires_grad or slots.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            from ... import train
            return train.slot_attention_with_grad(self, inputs, slots)   # one autograd node (row N1)
        return self._run(inputs, slots, False)[0]

    @property
    def dtype(self):
        return self.project_k.weight.dtype

    @property
    def device(self):
        return self.project_k.weight.device


class StoSAVi(BaseModel):
    """SAVi with stochastic kernels (`kld_method='none'` gives plain SAVi) -- the API of the reference's StoSAVi
    (savi.py:113-546): same constructor keywords, attributes, output dictionaries and state-dict keys."""

    DEFAULTS = dict(
        slot_dict=dict(num_slots=7, slot_size=128, slot_mlp_size=256, num_iterations=2, kernel_mlp=True),
        enc_dict=dict(enc_channels=(3, 64, 64, 64, 64), enc_ks=5, enc_out_channels=128, enc_norm=''),
        dec_dict=dict(dec_channels=(128, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5, dec_norm=''),
        pred_dict=dict(pred_type='transformer', pred_rnn=True, pred_norm_first=True, pred_num_layers=2, pred_num_heads=4,
                       pred_ffn_dim=512, pred_sg_every=None),
        loss_dict=dict(use_post_recon_loss=True, kld_method='var-0.01'),
    )

    def __init__(self, resolution, clip_len, slot_dict=None, enc_dict=None, dec_dict=None, pred_dict=None, loss_dict=None,
                 eps=1e-6):
        super().__init__()
        self.resolution, self.clip_len, self.eps = resolution, clip_len, eps
        given = dict(slot_dict=slot_dict, enc_dict=enc_dict, dec_dict=dec_dict, pred_dict=pred_dict, loss_dict=loss_dict)
        for name, value in given.items():
            setattr(self, name, value if value is not None else dict(self.DEFAULTS[name]))
        # registration order = state-dict order: slot attention side, CNN encoder, decoder, predictor
        self._build_slot_attention()
        self._build_encoder()
        self._build_decoder()
        self._build_predictor()
        self._build_loss()
        self.testing = False   # extraction mode: forward returns after the encode (savi.py:174-175, 487-488)

    # ---- parameter containers (nesting = checkpoint keys; see host/containers.py) -------------------------------
    def _build_slot_attention(self):
        sd = self.slot_dict
        self.enc_out_channels = self.enc_dict['enc_out_channels']
        self.num_slots, self.slot_size = sd['num_slots'], sd['slot_size']
        self.slot_mlp_size, self.num_iterations = sd['slot_mlp_size'], sd['num_iterations']
        D = self.slot_size
        self.init_latents = nn.Parameter(nn.init.normal_(torch.empty(1, self.num_slots, D)))
        # slots -> (mu | log_var) of the next kernels; one Linear, or Linear-LN-ReLU-Linear when `kernel_mlp` (default)
        self.kernel_dist_layer = (containers.dense_stack([D, 2 * D, 2 * D], norm_after_first=True)
                                  if sd.get('kernel_mlp', True) else containers.dense_stack([D, 2 * D]))
        # never used in the forward; the reference keeps it so that old checkpoints load (savi.py:202-209)
        self.prior_slot_layer = containers.dense_stack([D, D, D], norm_after_first=True)
        self.slot_attention = SlotAttention(in_features=self.enc_out_channels, num_iterations=self.num_iterations,
                                            num_slots=self.num_slots, slot_size=D, mlp_hidden_size=self.slot_mlp_size,
                                            eps=self.eps)

    def _build_encoder(self):
        ed = self.enc_dict
        self.enc_channels, self.enc_ks, self.enc_norm = list(ed['enc_channels']), ed['enc_ks'], ed['enc_norm']
        self.visual_resolution = (64, 64)   # 64x64 inputs at stride 1, 128x128 at stride 2 (savi.py:226,236)
        self.visual_channels = self.enc_channels[-1]
        self.encoder = containers.conv_stack(self.enc_channels, self.enc_ks, self.enc_norm,
                                             first_stride=2 if self.resolution[0] == 128 else 1)
        self.encoder_pos_embedding = SoftPositionEmbed(self.visual_channels, self.visual_resolution)
        self.encoder_out_layer = containers.dense_stack([self.visual_channels, self.enc_out_channels, self.enc_out_channels],
                                                        norm_first=True)

    def _build_decoder(self):
        """Spatial-broadcast decoder parameters (savi.py:252-293); the arithmetic is sf_savi_decode_f32."""
        dd = self.dec_dict
        self.dec_channels, self.dec_resolution = dd['dec_channels'], dd['dec_resolution']
        self.dec_ks, self.dec_norm = dd['dec_ks'], dd['dec_norm']
        assert self.dec_channels[0] == self.slot_size, 'wrong in_channels for Decoder'
        self.decoder, reached = containers.deconv_stack(self.dec_channels, self.dec_ks, self.dec_norm, self.dec_resolution[0],
                                                        self.resolution[0])
        assert_shape(self.resolution, (reached, reached),
                     message='Output shape of decoder did not match input resolution. Try changing `decoder_resolution`.')
        self.decoder_pos_embedding = SoftPositionEmbed(self.slot_size, self.dec_resolution)

    def _build_predictor(self):
        pd = self.pred_dict
        if pd.get('pred_type', 'transformer') == 'mlp':
            core = ResidualMLPPredictor([self.slot_size, self.slot_size * 2, self.slot_size], norm_first=pd['pred_norm_first'])
        else:
            core = TransformerPredictor(self.slot_size, pd['pred_num_layers'], pd['pred_num_heads'], pd['pred_ffn_dim'],
                                        norm_first=pd['pred_norm_first'])
        self.predictor = core if not pd['pred_rnn'] else RNNPredictorWrapper(
            core, self.slot_size, self.slot_mlp_size, num_layers=1, rnn_cell='LSTM', sg_every=pd['pred_sg_every'])

    def _build_loss(self):
        """kld_method: 'none' (deterministic kernels) or 'var-<v>' / 'var' (prior variance v, default 1)."""
        self.use_post_recon_loss = self.loss_dict['use_post_recon_loss']
        assert self.use_post_recon_loss
        method, _, var = self.loss_dict['kld_method'].partition('-')
        self.kld_log_var = math.log(float(var)) if var else math.log(1.)
        self.kld_method = method
        assert self.kld_method in ['var', 'none']

    # ---- hot path -------------------------------------------------------------------------------------------------
    def _draw_noise(self, B, T, device):
        """One randn per frame, also at eval, exactly like the reference's `_sample_dist` (savi.py:355-365): a seeded
        run consumes the generator identically."""
        if self.kld_method == 'none':
            return None
        return torch.stack([torch.randn(B, self.num_slots, self.slot_size, device=device) for _ in range(T)], 1)

    def _encode_with_grad(self, img, prev_slots, noise):
        """The same recurrence (savi.py:379-416) as a chain of autograd nodes on the HIP library (row N1): one node for
        the image encoder of all frames, then per frame predictor -> kernel distribution -> sample -> Slot Attention.
        Only the tiny elementwise glue on [B,N,D] tensors (sampling, residual add) runs as torch ops."""
        from ... import train
        B, T = img.shape[:2]
        feats = train.features_with_grad(self, img.transpose(0, 1).flatten(0, 1))   # time-major: feats[t*B:(t+1)*B] is contiguous
        feats = feats.unflatten(0, (T, B))
        kd_layers, D = getattr(self, 'kernel_dist_layer', None), self.slot_size   # STEVE has no kernel distribution
        dists, posts = [], []
        for t in range(T):
            if prev_slots is None:
                latents = self.init_latents.repeat(B, 1, 1)
            else:
                latents = self._predict_with_grad(prev_slots, train)
            if kd_layers is None:
                dist = kernels = latents
            else:
                if len(kd_layers) == 1:
                    dist = train.linear(latents, kd_layers[0])
                else:   # kernel_mlp: Linear -> LayerNorm -> ReLU -> Linear (savi.py:190-200)
                    dist = train.linear(torch.relu(train.layer_norm(train.linear(latents, kd_layers[0]), kd_layers[1])), kd_layers[3])
                kernels = dist[..., :D]
                if noise is not None:
                    kernels = kernels + noise[:, t] * torch.exp(0.5 * dist[..., D:])
            prev_slots = train.slot_attention_with_grad(self.slot_attention, feats[t], kernels.contiguous())
            dists.append(dist)
            posts.append(prev_slots)
        return torch.stack(dists, 1), torch.stack(posts, 1), None

    def _predict_with_grad(self, slots, train):
        """predictor(prev_slots) under autograd (predictor.py:20-135): residual MLP, or Transformer over the slots, each
        optionally followed by the one-step LSTM + projection of RNNPredictorWrapper (its state lives on the module and
        carries the graph from frame to frame; `sg_every` detaches it on schedule)."""
        pred = self.predictor
        rnn = pred if isinstance(pred, RNNPredictorWrapper) else None
        core = rnn.base_predictor if rnn is not None else pred
        if rnn is not None and rnn.sg_every is not None and rnn.step % rnn.sg_every == 0 and rnn.step > 0:
            slots = slots.detach()
            if rnn.hidden_state is not None:
                rnn.hidden_state = tuple(h.detach() for h in rnn.hidden_state)
        if isinstance(core, ResidualMLPPredictor):
            x = train.layer_norm(slots, core.ln)
            out = train.linear(train.linear(x, core.mlp[0], relu=True), core.mlp[2]) + (x if core.norm_first else slots)
        else:
            out = slots
            for layer in core.transformer_encoder.layers:
                out = train.transformer_encoder_layer(out, layer)
        if rnn is None:
            return out
        shape = out.shape
        state = None if rnn.hidden_state is None else tuple(h.reshape(-1, rnn.hidden_size) for h in rnn.hidden_state)
        h, state = train.lstm_step(out.reshape(-1, shape[-1]), state, rnn.rnn)
        rnn.hidden_state = tuple(t_.unsqueeze(0) for t_ in state)   # nn.LSTM's [1, B*N, H] layout
        rnn.step += 1
        return train.linear(h, rnn.out_projector).view(shape)

    def encode(self, img, prev_slots=None, noise=None):
        """img [B,T,3,H,W] -> (kernel_dist [B,T,N,2D], post_slots [B,T,N,D], None).  `noise` [B,T,N,D] injects the
        stochastic-kernel noise.  The third item is the reference's `encoder_out`, which nothing consumes
        (savi.py:478-485) and the fused engine does not materialise."""
        if self.kld_method == 'none':
            noise = None
        elif noise is None:
            noise = self._draw_noise(img.shape[0], img.shape[1], img.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self._encode_with_grad(img.float().contiguous(), prev_slots, noise)
        post, kdist, _ = engine.savi_encode(self, img, prev_slots=prev_slots, noise=noise)
        return kdist, post, None

    def _reset_rnn(self):
        self.predictor.reset()

    def forward(self, data_dict):
        """{'img' [B,T,3,H,W] (+ 'noise')} -> {'post_slots', 'kernel_dist', 'img'} (+ reconstructions unless `testing`).
        The engine walks the clip one time step at a time with O(1) activation memory in T, so the reference's
        OOM-probing temporal chunking (savi.py:421-463) is unnecessary: any T is one call, and the result equals every
        chunking because the predictor state is carried frame to frame."""
        img = data_dict['img']
        if not self.training:
            self.clip_len = max(self.clip_len, img.shape[1])
        return self._forward(img, None, noise=data_dict.get('noise', None))

    def _forward(self, img, prev_slots=None, noise=None):
        if prev_slots is None:
            self._reset_rnn()   # a new video: forget the predictor's recurrent state
        kernel_dist, post_slots, _ = self.encode(img, prev_slots=prev_slots, noise=noise)
        out = {'post_slots': post_slots, 'kernel_dist': kernel_dist, 'img': img}
        if self.testing or not self.use_post_recon_loss:
            return out
        B, T = img.shape[:2]
        recon, per_slot, masks, _ = self.decode(post_slots.flatten(0, 1))
        for name, t in (('post_recon_combined', recon), ('post_recons', per_slot), ('post_masks', masks)):
            out[name] = t.unflatten(0, (B, T))
        return out

    def decode(self, slots):
        """slots [F,N,D] -> (recon_combined [F,3,H,W], recons [F,N,3,H,W], masks [F,N,1,H,W], slots): savi.py:504-525 on
        the HIP decoder engine."""
        recon_combined, recons, masks = engine.savi_decode(self, slots)
        return recon_combined, recons, masks, slots

    def _kld_loss(self, prior_dist, post_slots):
        if self.kld_method == 'none':
            return torch.tensor(0.).type_as(prior_dist)
        return losses.kernel_kld(prior_dist, self.slot_size, self.kld_log_var)

    def calc_train_loss(self, data_dict, out_dict):
        terms = {'kld_loss': self._kld_loss(out_dict['kernel_dist'], out_dict['post_slots'])}
        if self.use_post_recon_loss:
            terms['post_recon_loss'] = losses.image_recon_loss(out_dict['post_recon_combined'], out_dict['img'])
        return terms

    @property
    def dtype(self):
        return self.slot_attention.dtype

    @property
    def device(self):
        return self.slot_attention.device
