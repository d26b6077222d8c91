"""STEVE slot-extraction side on the MI355X engine (reference: slotformer/base_slots/models/steve.py).

In scope: SlotAttentionWMask and STEVE.encode -- slots plus the
last-iteration attention as segmentation masks (bilinearly resized to the input resolution
in eval).  The dVAE tokenizer and the slot-conditioned Transformer decoder (training targets /
image decoding) are rows N2/out-of-scope; their checkpoint keys (`dvae.*`, `trans_decoder.*`)
are accepted and ignored when loading.
"""
import torch
from torch import nn

from ...nerv_compat import BaseModel
from ... import engine, ops
from .savi import SlotAttention, StoSAVi

_IGNORED_PREFIXES = ('dvae.', 'trans_decoder.')


class SlotAttentionWMask(SlotAttention):
    """Slot Attention that also returns the last attention map as seg mask (steve.py:13-73)."""

    def forward(self, inputs, slots):
        slots, mask = self._run(inputs, slots, True)
        return slots, mask  # [B,N,D], [B,N,HW]


class STEVE(StoSAVi):

    def __init__(
        self,
        resolution,
        clip_len,
        slot_dict=dict(num_slots=7, slot_size=128, slot_mlp_size=256, num_iterations=2),
        dvae_dict=dict(down_factor=4, vocab_size=4096, dvae_ckp_path=''),
        enc_dict=dict(enc_channels=(3, 64, 64, 64, 64), enc_ks=5, enc_out_channels=128, enc_norm=''),
        dec_dict=dict(dec_type='slate', dec_num_layers=4, dec_num_heads=4, dec_d_model=128),
        pred_dict=dict(pred_rnn=True, pred_norm_first=True, pred_num_layers=2, pred_num_heads=4,
                       pred_ffn_dim=512, pred_sg_every=None),
        loss_dict=dict(use_img_recon_loss=False),
        eps=1e-6,
    ):
        BaseModel.__init__(self)
        self.resolution = resolution
        self.clip_len = clip_len
        self.eps = eps
        self.slot_dict = slot_dict
        self.dvae_dict = dvae_dict
        self.enc_dict = enc_dict
        self.dec_dict = dec_dict
        self.pred_dict = pred_dict
        self.loss_dict = loss_dict

        self._build_slot_attention()
        self._build_encoder()
        self._build_predictor()
        self._build_loss()
        self.testing = False
        self._register_load_state_dict_pre_hook(self._drop_decoder_keys)

    @staticmethod
    def _drop_decoder_keys(state_dict, prefix, *args):
        for k in [k for k in state_dict if k[len(prefix):].startswith(_IGNORED_PREFIXES)]:
            del state_dict[k]

    def _build_slot_attention(self):
        self.enc_out_channels = self.enc_dict['enc_out_channels']
        self.num_slots = self.slot_dict['num_slots']
        self.slot_size = self.slot_dict['slot_size']
        self.slot_mlp_size = self.slot_dict['slot_mlp_size']
        self.num_iterations = self.slot_dict['num_iterations']
        self.init_latents = nn.Parameter(nn.init.normal_(torch.empty(1, self.num_slots, self.slot_size)))
        self.slot_attention = SlotAttentionWMask(
            in_features=self.enc_out_channels,
            num_iterations=self.num_iterations,
            num_slots=self.num_slots,
            slot_size=self.slot_size,
            mlp_hidden_size=self.slot_mlp_size,
            eps=self.eps,
        )

    def _build_loss(self):
        self.use_img_recon_loss = self.loss_dict['use_img_recon_loss']
        self.kld_method = 'none'

    def encode(self, img, prev_slots=None):
        """steve.py:198-240 -> (slots [B,T,N,D], masks [B,T,N,H,W], encoder_out=None)."""
        B, T = img.shape[:2]
        slots, _, attn = engine.savi_encode(self, img, prev_slots=prev_slots, noise=None, want_attn=True)
        masks = attn.view(B, T, self.num_slots, *self.visual_resolution)
        if not self.training and tuple(self.visual_resolution) != tuple(self.resolution):
            masks = ops.bilinear_resize(masks, tuple(self.resolution))
        return slots, masks, None

    def forward(self, data_dict):
        return self._forward(data_dict['img'], img_token_id=data_dict.get('token_id', None))

    def _forward(self, img, img_token_id=None, prev_slots=None):
        if prev_slots is None:
            self._reset_rnn()
        slots, masks, _ = self.encode(img, prev_slots)
        out_dict = {'slots': slots, 'masks': masks}
        if self.testing:
            return out_dict
        raise NotImplementedError(
            'STEVE token prediction / image decoding (dVAE + Transformer decoder, steve.py:306-337) is outside '
            'the slot-extraction hot path; set model.testing = True')
