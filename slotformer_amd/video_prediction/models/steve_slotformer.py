"""STEVESlotFormer on the MI355X engine (reference: slotformer/video_prediction/models/steve_slotformer.py).

SlotFormer rollout on STEVE slots; images are decoded by greedy generation of the dVAE token grid with the frozen
slot-conditioned Transformer decoder, then dVAE detokenisation.  Inference only."""
import copy

import torch
import torch.nn.functional as F

from ...nerv_compat import BaseModel
from ... import ops
from ...base_slots.models.steve import STEVE
from .slotformer import SlotFormer


class STEVESlotFormer(SlotFormer):

    def __init__(
            self,
            resolution,
            clip_len,
            slot_dict=dict(num_slots=6, slot_size=192),
            dvae_dict=dict(down_factor=4, vocab_size=4096, dvae_ckp_path=''),
            dec_dict=dict(dec_num_layers=4, dec_num_heads=4, dec_d_model=192, dec_ckp_path=''),
            rollout_dict=dict(num_slots=6, slot_size=192, history_len=6, t_pe='sin', slots_pe='', d_model=192,
                              num_layers=4, num_heads=8, ffn_dim=192 * 4, norm_first=True),
            loss_dict=dict(rollout_len=6, use_img_recon_loss=False),
            eps=1e-6,
    ):
        self.dvae_dict = dvae_dict
        super().__init__(resolution=resolution, clip_len=clip_len, slot_dict=slot_dict, dec_dict=dec_dict,
                         rollout_dict=rollout_dict, loss_dict=loss_dict, eps=eps)

    def _build_dvae(self):
        STEVE._build_dvae(self)

    def _build_decoder(self):
        """steve_slotformer.py:66-84: dVAE first, then the STEVE Transformer decoder under the name `decoder`, loaded
        from the `trans_decoder.*` keys of a STEVE checkpoint and frozen."""
        self._build_dvae()
        STEVE._build_decoder(self)
        self.decoder = copy.deepcopy(self.trans_decoder)
        del self.trans_decoder
        ckp_path = self.dec_dict['dec_ckp_path']
        assert ckp_path, 'Please provide pretrained Transformer decoder weight'
        w = torch.load(ckp_path, map_location='cpu')['state_dict']
        w = {k[14:]: v for k, v in w.items() if k.startswith('trans_decoder.')}
        self.decoder.load_state_dict(w)
        for p in self.decoder.parameters():
            p.requires_grad = False
        self.decoder.eval()

    def decode(self, slots, gumbel=None):
        """steve_slotformer.py:86-103: slots [B,N,D] -> (soft_recon, hard_recon) [B,3,H,W].  `gumbel` [B,V,h,w] injects the
        Gumbel noise of the soft relaxation (the reference draws it internally: -(Exp(1) + tiny).log())."""
        # K/V-cached greedy generation: same tokens / logits as decoder.generate(sample=False), O(P) instead of O(P^2) work
        _, logits = self.decoder.generate_cached(slots, steps=self.num_patches)   # [B,P,V] on the CPU
        logits = logits.to(slots.device).contiguous()
        B, P, V = logits.shape
        if gumbel is None:
            eps = torch.finfo(logits.dtype).tiny
            g_rows = -(torch.empty_like(logits).exponential_() + eps).log()
        else:
            g_rows = gumbel.to(logits.device).flatten(2, 3).transpose(1, 2).contiguous()   # [B,V,h,w] -> [B,P,V]
        z = ops.softmax_rows(logits, g_rows, 1.0 / 0.1)                                      # gumbel_softmax(log_softmax, 0.1)
        z = z.transpose(2, 1).unflatten(-1, (self.h, self.w)).contiguous()
        soft_recon = self.dvae.detokenize(z)
        idx = ops.argmax_rows(logits)                                                        # make_one_hot(logits, dim=1)
        z_hard = torch.zeros(B, V, P, device=logits.device, dtype=logits.dtype).scatter_(1, idx.unsqueeze(1), 1.)
        hard_recon = self.dvae.detokenize(z_hard.unflatten(-1, (self.h, self.w)).contiguous())
        return soft_recon, hard_recon

    def rollout(self, past_slots, pred_len, decode=False, with_gt=True):
        """steve_slotformer.py:105-109 (never decodes)."""
        return self.rollouter(past_slots[:, -self.history_len:], pred_len)

    def forward(self, data_dict):
        """steve_slotformer.py:111-148."""
        slots = data_dict['slots']
        assert self.rollout_len + self.history_len == slots.shape[1], \
            f'wrong SlotFormer training length {slots.shape[1]}'
        past_slots = slots[:, :self.history_len]
        gt_slots = slots[:, self.history_len:]
        pred_slots = self.rollout(past_slots, self.rollout_len)
        out_dict = {'gt_slots': gt_slots, 'pred_slots': pred_slots}
        if self.use_img_recon_loss:   # the token reconstruction loss of STEVE
            if 'token_id' in data_dict:
                gt_token_id = data_dict['token_id']
            else:
                gt_img = data_dict['img'][:, self.history_len:]
                gt_token_id = self.dvae.tokenize(gt_img, one_hot=False).flatten(2, 3)
            h, w = self.h, self.w
            target_token_id = gt_token_id.flatten(0, 1).long().contiguous()
            in_slots = pred_slots.flatten(0, 1)
            in_token_id = target_token_id[:, :-1].contiguous()
            pred_token_id = self.decoder(in_slots, in_token_id)[:, -(h * w):]
            out_dict.update({'pred_token_id': pred_token_id, 'target_token_id': target_token_id})
        return out_dict

    def calc_train_loss(self, data_dict, out_dict):
        """steve_slotformer.py:150-161 (values only)."""
        loss_dict = {'slot_recon_loss': F.mse_loss(out_dict['pred_slots'], out_dict['gt_slots'])}
        if self.use_img_recon_loss:
            pred = out_dict['pred_token_id'].flatten(0, 1).contiguous()
            target = out_dict['target_token_id'].flatten(0, 1).contiguous()
            loss_dict['img_recon_loss'] = ops.cross_entropy(pred, target)
        return loss_dict

    def train(self, mode=True):
        BaseModel.train(self, mode)
        self.dvae.eval()
        self.decoder.eval()
        return self
