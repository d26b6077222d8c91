"""STEVE slot-extraction side on the MI355X engine (reference: slotformer/base_slots/models/steve.py).

SlotAttentionWMask and STEVE.encode -- slots plus the last-iteration attention as segmentation masks (bilinearly
resized to the input resolution in eval) -- and the image side: dVAE token targets, the
teacher-forced slot-conditioned Transformer decoder and the token cross-entropy of `_forward` / `calc_train_loss`
(forward values only: the engine is inference-only, backward belongs to row N1).
"""
import torch
from torch import nn

from ...nerv_compat import BaseModel
from ... import engine, ops
from .dVAE import dVAE
from .savi import SlotAttention, StoSAVi
from .steve_transformer import STEVETransformerDecoder


class SlotAttentionWMask(SlotAttention):
    """Slot Attention that also returns the last attention map as seg mask (steve.py:13-73)."""

    def forward(self, inputs, slots):
        slots, mask = self._run(inputs, slots, True)
        return slots, mask  # [B,N,D], [B,N,HW]


class STEVE(StoSAVi):
    """STEVE (steve.py:76-351): SAVi's encoder side with mask-returning Slot Attention, plus the frozen dVAE tokenizer and
    the slot-conditioned Transformer decoder that predicts its tokens."""

    DEFAULTS = dict(
        slot_dict=dict(num_slots=7, slot_size=128, slot_mlp_size=256, num_iterations=2),
        dvae_dict=dict(down_factor=4, vocab_size=4096, dvae_ckp_path=''),
        enc_dict=dict(enc_channels=(3, 64, 64, 64, 64), enc_ks=5, enc_out_channels=128, enc_norm=''),
        dec_dict=dict(dec_type='slate', dec_num_layers=4, dec_num_heads=4, dec_d_model=128),
        pred_dict=dict(pred_rnn=True, pred_norm_first=True, pred_num_layers=2, pred_num_heads=4, pred_ffn_dim=512,
                       pred_sg_every=None),
        loss_dict=dict(use_img_recon_loss=False),
    )

    def __init__(self, resolution, clip_len, slot_dict=None, dvae_dict=None, enc_dict=None, dec_dict=None, pred_dict=None,
                 loss_dict=None, eps=1e-6):
        BaseModel.__init__(self)
        self.resolution, self.clip_len, self.eps = resolution, clip_len, eps
        given = dict(slot_dict=slot_dict, dvae_dict=dvae_dict, enc_dict=enc_dict, dec_dict=dec_dict, pred_dict=pred_dict,
                     loss_dict=loss_dict)
        for name, value in given.items():
            setattr(self, name, value if value is not None else dict(self.DEFAULTS[name]))
        # registration order = state-dict order of the reference: slot attention, dvae, encoder, trans_decoder, predictor
        self._build_slot_attention()
        self._build_dvae()
        self._build_encoder()
        self._build_decoder()
        self._build_predictor()
        self._build_loss()
        self.testing = False

    def _build_slot_attention(self):
        sd = self.slot_dict
        self.enc_out_channels = self.enc_dict['enc_out_channels']
        self.num_slots, self.slot_size = sd['num_slots'], sd['slot_size']
        self.slot_mlp_size, self.num_iterations = sd['slot_mlp_size'], sd['num_iterations']
        # STEVE's slots start from learned embeddings and there are no stochastic kernels (no kernel_dist_layer)
        self.init_latents = nn.Parameter(nn.init.normal_(torch.empty(1, self.num_slots, self.slot_size)))
        self.slot_attention = SlotAttentionWMask(in_features=self.enc_out_channels, num_iterations=self.num_iterations,
                                                 num_slots=self.num_slots, slot_size=self.slot_size,
                                                 mlp_hidden_size=self.slot_mlp_size, eps=self.eps)

    def _build_dvae(self):
        """steve.py:148-160.  The reference asserts a checkpoint path; an empty path here leaves the tokenizer at its
        initial weights (they normally arrive with the STEVE checkpoint's `dvae.*` keys)."""
        self.vocab_size = self.dvae_dict['vocab_size']
        self.down_factor = self.dvae_dict['down_factor']
        self.dvae = dVAE(vocab_size=self.vocab_size, img_channels=3)
        ckp_path = self.dvae_dict.get('dvae_ckp_path', '')
        if ckp_path:
            ckp = torch.load(ckp_path, map_location='cpu')
            self.dvae.load_state_dict(ckp['state_dict'])
        for p in self.dvae.parameters():
            p.requires_grad = False
        self.dvae.eval()

    def _build_decoder(self):
        """steve.py:162-176: GPT-style causal Transformer decoder over the h*w image tokens, conditioned on the slots."""
        H, W = self.resolution
        self.h, self.w = H // self.down_factor, W // self.down_factor
        self.num_patches = self.h * self.w
        self.trans_decoder = STEVETransformerDecoder(
            vocab_size=self.vocab_size,
            d_model=self.dec_dict['dec_d_model'],
            n_head=self.dec_dict['dec_num_heads'],
            max_len=self.num_patches - 1,
            num_slots=self.num_slots,
            num_layers=self.dec_dict['dec_num_layers'],
        )

    def _build_loss(self):
        self.use_img_recon_loss = self.loss_dict['use_img_recon_loss']
        self.kld_method = 'none'

    def encode(self, img, prev_slots=None):
        """steve.py:198-240 -> (slots [B,T,N,D], masks [B,T,N,H,W], encoder_out=None)."""
        B, T = img.shape[:2]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training (row N1): the slots come from the chain of autograd nodes shared with StoSAVi; the seg masks are
            # detached in the reference (steve.py:54-55), so they come from the inference engine in a no-grad pass
            state = None if not hasattr(self.predictor, 'hidden_state') else (self.predictor.step, self.predictor.hidden_state)
            with torch.no_grad():
                _, _, attn = engine.savi_encode(self, img, prev_slots=None if prev_slots is None else prev_slots.detach(), noise=None,
                                                want_attn=True)
            if state is not None:
                self.predictor.step, self.predictor.hidden_state = state
            _, slots, _ = self._encode_with_grad(img.float().contiguous(), prev_slots, None)
            return slots, attn.view(B, T, self.num_slots, *self.visual_resolution), None
        slots, _, attn = engine.savi_encode(self, img, prev_slots=prev_slots, noise=None, want_attn=True)
        masks = attn.view(B, T, self.num_slots, *self.visual_resolution)
        if not self.training and tuple(self.visual_resolution) != tuple(self.resolution):
            masks = ops.bilinear_resize(masks, tuple(self.resolution))
        return slots, masks, None

    def forward(self, data_dict):
        return self._forward(data_dict['img'], img_token_id=data_dict.get('token_id', None), gumbel=data_dict.get('gumbel', None))

    def _forward(self, img, img_token_id=None, prev_slots=None, gumbel=None):
        if prev_slots is None:
            self._reset_rnn()
        slots, masks, _ = self.encode(img, prev_slots)
        out_dict = {'slots': slots, 'masks': masks}
        if self.testing:
            return out_dict
        # token targets from the frozen dVAE, teacher-forced decoder logits (steve.py:306-322)
        if img_token_id is None:
            with torch.no_grad():   # the dVAE is frozen: the tokens are targets, not part of the graph (steve.py:306-311)
                img_token_id = self.dvae.tokenize(img, one_hot=False).flatten(2, 3)
        h, w = self.h, self.w
        target_token_id = img_token_id.flatten(0, 1).long().contiguous()   # [B*T, h*w]
        in_slots = slots.flatten(0, 1)
        in_token_id = target_token_id[:, :-1].contiguous()
        pred_token_id = self.trans_decoder(in_slots, in_token_id)
        if pred_token_id.shape[1] != h * w:   # (a no-op slice would still cost autograd a zero-filled copy of the logits)
            pred_token_id = pred_token_id[:, -(h * w):]
        out_dict.update({'pred_token_id': pred_token_id, 'target_token_id': target_token_id})
        if self.use_img_recon_loss:
            # steve.py:327-335: relaxed sample (tau 0.1) of the predicted token map, decoded by the frozen dVAE.  `gumbel`
            # [B*T,V,h,w] fixes the noise the reference draws inside gumbel_softmax.
            from ... import train
            logits = pred_token_id.reshape(-1, h, w, self.vocab_size)
            g = None   # no noise given: generated inside the softmax kernel
            if gumbel is not None:
                g = gumbel.reshape(-1, self.vocab_size, h, w).permute(0, 2, 3, 1).to(logits.device).float().contiguous()
            z = train.gumbel_softmax(logits, g, tau=0.1, hard=False)
            out_dict['gt_img'] = img.flatten(0, 1)
            out_dict['recon_img'] = self.dvae.decode_nhwc(z).permute(0, 3, 1, 2)
        return out_dict

    def calc_train_loss(self, data_dict, out_dict):
        """Token cross-entropy of steve.py:339-351; differentiable when the logits carry a graph (row N1)."""
        pred = out_dict['pred_token_id'].flatten(0, 1).contiguous()
        target = out_dict['target_token_id'].flatten(0, 1).contiguous()
        if pred.requires_grad:
            from ... import train
            loss_dict = {'token_recon_loss': train.token_cross_entropy(pred, target)}
        else:
            loss_dict = {'token_recon_loss': ops.cross_entropy(pred, target)}
        if self.use_img_recon_loss:
            from ...host import losses
            loss_dict['img_recon_loss'] = losses.image_recon_loss(out_dict['recon_img'], out_dict['gt_img'])
        return loss_dict

    def train(self, mode=True):
        """steve.py: the dVAE stays in eval mode."""
        super().train(mode)
        self.dvae.eval()
        return self
