"""SAVi / StoSAVi slot-extraction model on the MI355X engine.

Mirrors the public surface of the reference's slotformer/base_slots/models/savi.py
(constructor arguments, attribute names, output dicts, state-dict keys) so checkpoints and the
extraction scripts work unchanged.  The nn.Modules below only *hold parameters*; the forward
arithmetic runs in libslotformer_hip (sf_savi_encode_f32).
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

from ...nerv_compat import BaseModel, conv_norm_act, deconv_norm_act, deconv_out_shape
from ... import engine, ops
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
        self.mlp = nn.Sequential(
            nn.LayerNorm(self.slot_size),
            nn.Linear(self.slot_size, self.mlp_hidden_size),
            nn.ReLU(),
            nn.Linear(self.mlp_hidden_size, self.slot_size),
        )

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
        return self._run(inputs, slots, False)[0]

    @property
    def dtype(self):
        return self.project_k.weight.dtype

    @property
    def device(self):
        return self.project_k.weight.device


class StoSAVi(BaseModel):
    """SAVi with stochastic kernels (`kld_method='none'` makes it plain SAVi); reference savi.py:113-546."""

    def __init__(
        self,
        resolution,
        clip_len,
        slot_dict=dict(num_slots=7, slot_size=128, slot_mlp_size=256, num_iterations=2, kernel_mlp=True),
        enc_dict=dict(enc_channels=(3, 64, 64, 64, 64), enc_ks=5, enc_out_channels=128, enc_norm=''),
        dec_dict=dict(dec_channels=(128, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5, dec_norm=''),
        pred_dict=dict(pred_type='transformer', pred_rnn=True, pred_norm_first=True, pred_num_layers=2,
                       pred_num_heads=4, pred_ffn_dim=512, pred_sg_every=None),
        loss_dict=dict(use_post_recon_loss=True, kld_method='var-0.01'),
        eps=1e-6,
    ):
        super().__init__()
        self.resolution = resolution
        self.clip_len = clip_len
        self.eps = eps
        self.slot_dict = slot_dict
        self.enc_dict = enc_dict
        self.dec_dict = dec_dict
        self.pred_dict = pred_dict
        self.loss_dict = loss_dict

        self._build_slot_attention()
        self._build_encoder()
        self._build_decoder()
        self._build_predictor()
        self._build_loss()

        # extraction mode: return slots only (reference savi.py:174-175, 487-488)
        self.testing = False

    # ---- construction (same submodule names => same state-dict keys as the reference) ----------
    def _build_slot_attention(self):
        self.enc_out_channels = self.enc_dict['enc_out_channels']
        self.num_slots = self.slot_dict['num_slots']
        self.slot_size = self.slot_dict['slot_size']
        self.slot_mlp_size = self.slot_dict['slot_mlp_size']
        self.num_iterations = self.slot_dict['num_iterations']
        self.init_latents = nn.Parameter(nn.init.normal_(torch.empty(1, self.num_slots, self.slot_size)))
        if self.slot_dict.get('kernel_mlp', True):
            self.kernel_dist_layer = nn.Sequential(
                nn.Linear(self.slot_size, self.slot_size * 2),
                nn.LayerNorm(self.slot_size * 2),
                nn.ReLU(),
                nn.Linear(self.slot_size * 2, self.slot_size * 2),
            )
        else:
            self.kernel_dist_layer = nn.Sequential(nn.Linear(self.slot_size, self.slot_size * 2), )
        # dead weights the reference keeps for checkpoint compatibility (savi.py:202-209)
        self.prior_slot_layer = nn.Sequential(
            nn.Linear(self.slot_size, self.slot_size),
            nn.LayerNorm(self.slot_size),
            nn.ReLU(),
            nn.Linear(self.slot_size, self.slot_size),
        )
        self.slot_attention = SlotAttention(
            in_features=self.enc_out_channels,
            num_iterations=self.num_iterations,
            num_slots=self.num_slots,
            slot_size=self.slot_size,
            mlp_hidden_size=self.slot_mlp_size,
            eps=self.eps,
        )

    def _build_encoder(self):
        self.enc_channels = list(self.enc_dict['enc_channels'])
        self.enc_ks = self.enc_dict['enc_ks']
        self.enc_norm = self.enc_dict['enc_norm']
        self.visual_resolution = (64, 64)
        self.visual_channels = self.enc_channels[-1]
        n = len(self.enc_channels) - 1
        self.encoder = nn.Sequential(*[
            conv_norm_act(
                self.enc_channels[i],
                self.enc_channels[i + 1],
                kernel_size=self.enc_ks,
                stride=2 if (i == 0 and self.resolution[0] == 128) else 1,
                norm=self.enc_norm,
                act='relu' if i != (n - 1) else '') for i in range(n)
        ])
        self.encoder_pos_embedding = SoftPositionEmbed(self.visual_channels, self.visual_resolution)
        self.encoder_out_layer = nn.Sequential(
            nn.LayerNorm(self.visual_channels),
            nn.Linear(self.visual_channels, self.enc_out_channels),
            nn.ReLU(),
            nn.Linear(self.enc_out_channels, self.enc_out_channels),
        )

    def _build_decoder(self):
        """Spatial-broadcast decoder parameters (savi.py:252-293); arithmetic in sf_savi_decode_f32."""
        self.dec_channels = self.dec_dict['dec_channels']
        self.dec_resolution = self.dec_dict['dec_resolution']
        self.dec_ks = self.dec_dict['dec_ks']
        self.dec_norm = self.dec_dict['dec_norm']
        assert self.dec_channels[0] == self.slot_size, 'wrong in_channels for Decoder'
        modules = []
        out_size = self.dec_resolution[0]
        stride = 2
        for i in range(len(self.dec_channels) - 1):
            if out_size == self.resolution[0]:
                stride = 1
            modules.append(
                deconv_norm_act(self.dec_channels[i], self.dec_channels[i + 1], kernel_size=self.dec_ks,
                                stride=stride, norm=self.dec_norm, act='relu'))
            out_size = deconv_out_shape(out_size, stride, self.dec_ks // 2, self.dec_ks, stride - 1)
        assert_shape(self.resolution, (out_size, out_size),
                     message="Output shape of decoder did not match input resolution. "
                     "Try changing `decoder_resolution`.")
        modules.append(nn.Conv2d(self.dec_channels[-1], 4, kernel_size=1, stride=1, padding=0))
        self.decoder = nn.Sequential(*modules)
        self.decoder_pos_embedding = SoftPositionEmbed(self.slot_size, self.dec_resolution)

    def _build_predictor(self):
        pred_type = self.pred_dict.get('pred_type', 'transformer')
        if pred_type == 'mlp':
            self.predictor = ResidualMLPPredictor(
                [self.slot_size, self.slot_size * 2, self.slot_size],
                norm_first=self.pred_dict['pred_norm_first'],
            )
        else:
            self.predictor = TransformerPredictor(
                self.slot_size,
                self.pred_dict['pred_num_layers'],
                self.pred_dict['pred_num_heads'],
                self.pred_dict['pred_ffn_dim'],
                norm_first=self.pred_dict['pred_norm_first'],
            )
        if self.pred_dict['pred_rnn']:
            self.predictor = RNNPredictorWrapper(
                self.predictor,
                self.slot_size,
                self.slot_mlp_size,
                num_layers=1,
                rnn_cell='LSTM',
                sg_every=self.pred_dict['pred_sg_every'],
            )

    def _build_loss(self):
        self.use_post_recon_loss = self.loss_dict['use_post_recon_loss']
        assert self.use_post_recon_loss
        kld_method = self.loss_dict['kld_method']
        if '-' in kld_method:
            kld_method, kld_var = kld_method.split('-')
            self.kld_log_var = math.log(float(kld_var))
        else:
            self.kld_log_var = math.log(1.)
        self.kld_method = kld_method
        assert self.kld_method in ['var', 'none']

    # ---- hot path ------------------------------------------------------------------------------
    def _draw_noise(self, B, T, device):
        """The reference draws randn_like(mu) once per frame, also at eval (savi.py:355-365);
        same call pattern here so a seeded run consumes the generator identically."""
        if self.kld_method == 'none':
            return None
        return torch.stack([torch.randn(B, self.num_slots, self.slot_size, device=device) for _ in range(T)], 1)

    def encode(self, img, prev_slots=None, noise=None):
        """img [B,T,3,H,W] -> (kernel_dist [B,T,N,2D], post_slots [B,T,N,D], encoder_out).

        `noise` ([B,T,N,D]) optionally injects the stochastic-kernel noise.  `encoder_out` is
        not materialised by the fused engine (it is never consumed, savi.py:478-485) -> None.
        """
        B, T = img.shape[:2]
        if noise is None:
            noise = self._draw_noise(B, T, img.device)
        elif self.kld_method == 'none':
            noise = None
        post, kdist, _ = engine.savi_encode(self, img, prev_slots=prev_slots, noise=noise)
        return kdist, post, None

    def _reset_rnn(self):
        self.predictor.reset()

    def forward(self, data_dict):
        """Reference savi.py:421-463.  The engine walks the clip one time step at a time with O(1)
        activation memory in T, so the reference's OOM-probing temporal chunking is not needed:
        any T goes through one call, with results identical to every chunking (the predictor
        state is carried frame to frame)."""
        img = data_dict['img']
        T = img.shape[1]
        self.clip_len = max(self.clip_len, T) if not self.training else self.clip_len
        return self._forward(img, None, noise=data_dict.get('noise', None))

    def _forward(self, img, prev_slots=None, noise=None):
        if prev_slots is None:
            self._reset_rnn()
        B, T = img.shape[:2]
        kernel_dist, post_slots, _ = self.encode(img, prev_slots=prev_slots, noise=noise)
        out_dict = {
            'post_slots': post_slots,  # [B, T, num_slots, C]
            'kernel_dist': kernel_dist,  # [B, T, num_slots, 2C]
            'img': img,  # [B, T, 3, H, W]
        }
        if self.testing:
            return out_dict
        if self.use_post_recon_loss:
            post_recon_img, post_recons, post_masks, _ = self.decode(post_slots.flatten(0, 1))
            post_dict = {
                'post_recon_combined': post_recon_img,  # [B*T, 3, H, W]
                'post_recons': post_recons,  # [B*T, num_slots, 3, H, W]
                'post_masks': post_masks,  # [B*T, num_slots, 1, H, W]
            }
            out_dict.update({k: v.unflatten(0, (B, T)) for k, v in post_dict.items()})
        return out_dict

    def decode(self, slots):
        """slots [F,N,D] -> (recon_combined [F,3,H,W], recons [F,N,3,H,W], masks [F,N,1,H,W], slots);
        reference savi.py:504-525, on the HIP decoder engine (sf_savi_decode_f32)."""
        recon_combined, recons, masks = engine.savi_decode(self, slots)
        return recon_combined, recons, masks, slots

    def _kld_loss(self, prior_dist, post_slots):
        """savi.py:337-353."""
        if self.kld_method == 'none':
            return torch.tensor(0.).type_as(prior_dist)
        assert prior_dist.shape[-1] == self.slot_size * 2
        mu1 = prior_dist[..., :self.slot_size]
        log_var1 = prior_dist[..., self.slot_size:]
        mu2 = mu1.detach().clone()
        log_var2 = torch.ones_like(log_var1).detach() * self.kld_log_var
        sigma1 = torch.exp(log_var1 * 0.5)
        sigma2 = torch.exp(log_var2 * 0.5)
        kld = torch.log(sigma2 / sigma1) + (torch.exp(log_var1) + (mu1 - mu2)**2) / (2. * torch.exp(log_var2)) - 0.5
        return kld.sum(-1).mean()

    def calc_train_loss(self, data_dict, out_dict):
        loss_dict = {'kld_loss': self._kld_loss(out_dict['kernel_dist'], out_dict['post_slots'])}
        if self.use_post_recon_loss:
            loss_dict['post_recon_loss'] = F.mse_loss(out_dict['post_recon_combined'], out_dict['img'])
        return loss_dict

    @property
    def dtype(self):
        return self.slot_attention.dtype

    @property
    def device(self):
        return self.slot_attention.device
