"""PHYRE variant: burn-in is one frame, the window grows to `cond_len` frames and then slides
(reference: slotformer/video_prediction/models/single_step_slotformer.py)."""
import torch

from .slotformer import build_pos_enc, SlotRollouter, SlotFormer


class SingleStepSlotRollouter(SlotRollouter):
    """single_step_slotformer.py:6-90.  The growing window and the tail-sliced PE
    (`enc_pe[:, -len:]`, :81) are handled inside sf_rollout_f32 (single_step = 1)."""

    def __init__(self, num_slots, slot_size, history_len, cond_len, t_pe='sin', slots_pe='', d_model=128,
                 num_layers=4, num_heads=8, ffn_dim=512, norm_first=True):
        super().__init__(num_slots=num_slots, slot_size=slot_size, history_len=history_len, t_pe=t_pe,
                         slots_pe=slots_pe, d_model=d_model, num_layers=num_layers, num_heads=num_heads,
                         ffn_dim=ffn_dim, norm_first=norm_first)
        assert self.history_len == 1, 'SingleStepSlotRollouter performs rollout using only initial frame'
        self.cond_len = cond_len
        self.num_cond_tokens = self.cond_len * self.num_slots
        self.enc_t_pe = build_pos_enc(t_pe, cond_len, d_model)


class SingleStepSlotFormer(SlotFormer):

    def _build_loss(self):
        super()._build_loss()
        self.use_cls_loss = False
        self.success_cls = None  # PHYRE success classifier: a downstream consumer, attached from outside

    def _build_rollouter(self):
        self.history_len = self.rollout_dict['history_len']  # 1
        self.rollouter = SingleStepSlotRollouter(**self.rollout_dict)

    def classify(self, slots, vid_len=None):
        assert not self.training
        return self.success_cls({'slots': slots, 'vid_len': vid_len})['logits']

    def forward(self, data_dict):
        """single_step_slotformer.py:119-129 (including its gt_slots/pred_slots concat quirk)."""
        out_dict = super().forward(data_dict)
        if not (self.use_cls_loss and self.success_cls is not None):
            return out_dict
        slots = torch.cat([out_dict['gt_slots'], out_dict['pred_slots']], dim=1)
        out_dict['logits'] = self.classify(slots, data_dict.get('vid_len', None))
        return out_dict
