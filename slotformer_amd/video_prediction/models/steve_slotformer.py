"""STEVESlotFormer on the MI355X engine (API of slotformer/video_prediction/models/steve_slotformer.py).

SlotFormer dynamics on STEVE slots.  Frames are rendered from slots by generating the dVAE token grid greedily with the
frozen slot-conditioned Transformer decoder (`sf_slate_generate_f32`: K/V cache, one C call) and detokenising it; the
optional "image" loss is the token cross-entropy of the teacher-forced decoder on the predicted slots.  Inference only."""
import copy

import torch

from ...nerv_compat import BaseModel
from ... import ops
from ...base_slots.models.steve import STEVE
from ...host import frozen
from .slotformer import SlotFormer

GUMBEL_TAU = 0.1   # temperature of the soft token map in decode (steve_slotformer.py:98)


class STEVESlotFormer(SlotFormer):

    def __init__(self, resolution, clip_len, slot_dict=None, dvae_dict=None, dec_dict=None, rollout_dict=None, loss_dict=None,
                 eps=1e-6):
        # the dVAE settings must exist before SlotFormer.__init__ builds the decoder
        self.dvae_dict = dvae_dict if dvae_dict is not None else dict(down_factor=4, vocab_size=4096, dvae_ckp_path='')
        if slot_dict is None:
            slot_dict = dict(num_slots=6, slot_size=192)
        if dec_dict is None:
            dec_dict = dict(dec_num_layers=4, dec_num_heads=4, dec_d_model=192, dec_ckp_path='')
        if rollout_dict is None:
            rollout_dict = dict(num_slots=6, slot_size=192, history_len=6, t_pe='sin', slots_pe='', d_model=192, num_layers=4,
                                num_heads=8, ffn_dim=192 * 4, norm_first=True)
        super().__init__(resolution=resolution, clip_len=clip_len, slot_dict=slot_dict, dec_dict=dec_dict,
                         rollout_dict=rollout_dict, loss_dict=loss_dict, eps=eps)

    # ---- construction: `dvae.*`, then the STEVE Transformer decoder under the name `decoder.*` --------------------------
    def _build_dvae(self):
        STEVE._build_dvae(self)

    def _build_decoder(self):
        self._build_dvae()
        STEVE._build_decoder(self)                       # creates self.trans_decoder (+ h, w, num_patches)
        self.decoder = copy.deepcopy(self.trans_decoder)
        del self.trans_decoder
        sd = frozen.checkpoint_state(self.dec_dict['dec_ckp_path'], 'Transformer decoder')
        frozen.load_prefixed(self.decoder, sd, 'trans_decoder.')
        frozen.freeze(self.decoder)

    # ---- rendering ------------------------------------------------------------------------------------------------
    def _token_map(self, rows):
        """[B, P, V] per-token rows -> [B, V, h, w] map."""
        return rows.transpose(2, 1).unflatten(-1, (self.h, self.w)).contiguous()

    def decode(self, slots, gumbel=None):
        """slots [B,N,D] -> (soft_recon, hard_recon), both [B,3,H,W] (steve_slotformer.py:86-103).  The soft image
        detokenises the Gumbel-softmax relaxation of the generated logits at tau = 0.1; `gumbel` [B,V,h,w] injects its
        noise (the reference draws -(Exp(1) + tiny).log() internally).  The hard image detokenises the one-hot argmax."""
        _, logits = self.decoder.generate_cached(slots, steps=self.num_patches)      # [B,P,V], returned on the CPU
        logits = logits.to(slots.device).contiguous()
        if gumbel is not None:
            noise = gumbel.to(logits.device).flatten(2, 3).transpose(1, 2).contiguous()
        else:
            noise = -(torch.empty_like(logits).exponential_() + torch.finfo(logits.dtype).tiny).log()
        soft = self.dvae.detokenize(self._token_map(ops.softmax_rows(logits, noise, 1.0 / GUMBEL_TAU)))
        picked = ops.argmax_rows(logits)                                             # [B,P]
        one_hot = torch.zeros_like(logits).scatter_(2, picked.unsqueeze(2), 1.)
        hard = self.dvae.detokenize(self._token_map(one_hot))
        return soft, hard

    def rollout(self, past_slots, pred_len, decode=False, with_gt=True):
        """Slots only: this model never renders inside rollout (steve_slotformer.py:105-109)."""
        return self.rollouter(past_slots[:, -self.history_len:], pred_len)

    def forward(self, data_dict):
        """{'slots' (+ 'img' or 'token_id' when the token loss is on)} -> {'pred_slots', 'gt_slots'} (+ token logits and
        targets for the predicted frames)."""
        slots = data_dict['slots']
        assert self.rollout_len + self.history_len == slots.shape[1], f'wrong SlotFormer training length {slots.shape[1]}'
        target = slots[:, self.history_len:]
        pred = self.rollout(slots[:, :self.history_len], self.rollout_len)
        out = {'gt_slots': target, 'pred_slots': pred}
        if not self.use_img_recon_loss:
            return out
        if 'token_id' in data_dict:
            ids = data_dict['token_id']
        else:
            with torch.no_grad():   # frozen tokenizer: the ids are targets, not part of the graph
                ids = self.dvae.tokenize(data_dict['img'][:, self.history_len:], one_hot=False).flatten(2, 3)
        ids = ids.flatten(0, 1).long().contiguous()                                   # [B*T, h*w]
        logits = self.decoder(pred.flatten(0, 1), ids[:, :-1].contiguous())
        out['pred_token_id'] = logits if logits.shape[1] == self.h * self.w else logits[:, -(self.h * self.w):]
        out['target_token_id'] = ids
        return out

    def calc_train_loss(self, data_dict, out_dict):
        """Values of steve_slotformer.py:150-161: plain slot MSE and, if enabled, the token cross-entropy (reported under
        the reference's name 'img_recon_loss')."""
        terms = {'slot_recon_loss': ((out_dict['pred_slots'] - out_dict['gt_slots'])**2).mean()}
        if self.use_img_recon_loss:
            logits = out_dict['pred_token_id'].flatten(0, 1).contiguous()
            target = out_dict['target_token_id'].flatten(0, 1).contiguous()
            if logits.requires_grad:   # training (row N1): the frozen decoder passes the gradient on to the predicted slots
                from ... import train
                terms['img_recon_loss'] = train.token_cross_entropy(logits, target)
            else:
                terms['img_recon_loss'] = ops.cross_entropy(logits, target)
        return terms

    def train(self, mode=True):
        BaseModel.train(self, mode)
        frozen.freeze(self.dvae, self.decoder)   # tokenizer and decoder stay frozen in eval mode
        return self
