"""SlotFormer rollout model on the MI355X engine.

Mirrors slotformer/video_prediction/models/slotformer.py (constructor arguments, attributes
poked from outside -- rollout_len, use_img_recon_loss, loss_decay_factor, testing -- output
dicts and state-dict keys).  The autoregressive loop runs in libslotformer_hip
(sf_rollout_f32): all slots live in one [B, T, N, C] device buffer, the Transformer window of
each step is a strided view of it and predictions are written in place.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...nerv_compat import BaseModel
from ... import engine
from ...base_slots.models import StoSAVi


def get_sin_pos_enc(seq_len, d_model):
    """[1, L, d] sinusoid PE; positions run L-1 .. 0, i.e. the newest frame is position 0;
    sin half then cos half (slotformer.py:10-16)."""
    inv_freq = 1. / (10000**(torch.arange(0.0, d_model, 2.0) / d_model))
    pos = torch.arange(seq_len - 1, -1, -1).type_as(inv_freq)
    ang = torch.outer(pos, inv_freq)
    return torch.cat([ang.sin(), ang.cos()], dim=-1).unsqueeze(0)


def build_pos_enc(pos_enc, input_len, d_model):
    """slotformer.py:19-32."""
    if not pos_enc:
        return None
    if pos_enc == 'learnable':
        return nn.Parameter(torch.zeros(1, input_len, d_model))
    if 'sin' in pos_enc:
        return nn.Parameter(get_sin_pos_enc(input_len, d_model), requires_grad=False)
    raise NotImplementedError(f'unsupported pos enc {pos_enc}')


class Rollouter(nn.Module):

    def burnin(self, x):
        pass

    def reset(self):
        pass


class SlotRollouter(Rollouter):
    """Transformer-encoder rollouter parameters (slotformer.py:48-134)."""

    def __init__(self, num_slots, slot_size, history_len, t_pe='sin', slots_pe='', d_model=128, num_layers=4,
                 num_heads=8, ffn_dim=512, norm_first=True):
        super().__init__()
        self.num_slots = num_slots
        self.history_len = history_len
        self.in_proj = nn.Linear(slot_size, d_model)
        enc_layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_heads, dim_feedforward=ffn_dim,
                                               norm_first=norm_first, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=enc_layer, num_layers=num_layers,
                                                         enable_nested_tensor=False)
        self.enc_t_pe = build_pos_enc(t_pe, history_len, d_model)
        self.enc_slots_pe = build_pos_enc(slots_pe, num_slots, d_model)
        self.out_proj = nn.Linear(d_model, slot_size)

    def _n_in(self):
        return self.history_len

    def forward(self, x, pred_len):
        """x [B, history_len, N, C] -> [B, pred_len, N, C]."""
        assert x.shape[1] == self.history_len, 'wrong burn-in steps'
        B, n_in, N, C = x.shape
        buf = torch.empty(B, n_in + pred_len, N, C, device=x.device, dtype=torch.float32)
        buf[:, :n_in] = x
        engine.rollout(self, buf, n_in, pred_len)
        return buf[:, n_in:]

    @property
    def dtype(self):
        return self.in_proj.weight.dtype

    @property
    def device(self):
        return self.in_proj.weight.device


class SlotFormer(BaseModel):
    """Transformer-based autoregressive dynamics model over slots (slotformer.py:137-343)."""

    def __init__(
            self,
            resolution,
            clip_len,
            slot_dict=dict(num_slots=7, slot_size=128),
            dec_dict=dict(dec_channels=(128, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5, dec_norm='',
                          dec_ckp_path=''),
            rollout_dict=dict(num_slots=7, slot_size=128, history_len=6, t_pe='sin', slots_pe='', d_model=128,
                              num_layers=4, num_heads=8, ffn_dim=512, norm_first=True),
            loss_dict=dict(rollout_len=6, use_img_recon_loss=False),
            eps=1e-6,
    ):
        super().__init__()
        self.resolution = resolution
        self.clip_len = clip_len
        self.eps = eps
        self.slot_dict = slot_dict
        self.dec_dict = dec_dict
        self.rollout_dict = rollout_dict
        self.loss_dict = loss_dict

        self._build_slot_attention()
        self._build_decoder()
        self._build_rollouter()
        self._build_loss()

        self.testing = False  # for compatibility
        self.loss_decay_factor = 1.  # temporal loss weighting

    def _build_slot_attention(self):
        self.num_slots = self.slot_dict['num_slots']
        self.slot_size = self.slot_dict['slot_size']

    def _build_decoder(self):
        """Frozen SAVi decoder copy (slotformer.py:196-218): same parameters, loaded from
        `dec_ckp_path` by key prefix."""
        StoSAVi._build_decoder(self)
        ckp_path = self.dec_dict['dec_ckp_path']
        assert ckp_path, 'Please provide pretrained decoder weight'
        w = torch.load(ckp_path, map_location='cpu')['state_dict']
        self.decoder.load_state_dict({k[len('decoder.'):]: v for k, v in w.items() if k.startswith('decoder.')})
        self.decoder_pos_embedding.load_state_dict({
            k[len('decoder_pos_embedding.'):]: v
            for k, v in w.items() if k.startswith('decoder_pos_embedding.')
        })
        for p in list(self.decoder.parameters()) + list(self.decoder_pos_embedding.parameters()):
            p.requires_grad = False
        self.decoder.eval()
        self.decoder_pos_embedding.eval()

    def _build_rollouter(self):
        self.history_len = self.rollout_dict['history_len']
        self.rollouter = SlotRollouter(**self.rollout_dict)

    def _build_loss(self):
        self.rollout_len = self.loss_dict['rollout_len']
        self.use_img_recon_loss = self.loss_dict['use_img_recon_loss']

    def decode(self, slots):
        return StoSAVi.decode(self, slots)

    def rollout(self, past_slots, pred_len, decode=False, with_gt=True):
        """slotformer.py:236-261."""
        B = past_slots.shape[0]
        pred_slots = self.rollouter(past_slots[:, -self.history_len:], pred_len)
        if decode:
            if with_gt:
                T = pred_len + past_slots.shape[1]
                slots = torch.cat([past_slots, pred_slots], dim=1)
            else:
                T = pred_len
                slots = pred_slots
            recon_img, recons, masks, _ = self.decode(slots.flatten(0, 1))
            out_dict = {'recon_combined': recon_img, 'recons': recons, 'masks': masks}
            out_dict = {k: v.unflatten(0, (B, T)) for k, v in out_dict.items()}
            out_dict['slots'] = slots
            return out_dict
        return pred_slots

    def forward(self, data_dict):
        """slotformer.py:263-282."""
        slots = data_dict['slots']  # [B, T, N, C]
        assert self.rollout_len + self.history_len == slots.shape[1], \
            f'wrong SlotFormer training length {slots.shape[1]}'
        past_slots = slots[:, :self.history_len]
        gt_slots = slots[:, self.history_len:]
        if self.use_img_recon_loss:
            out_dict = self.rollout(past_slots, self.rollout_len, decode=True, with_gt=False)
            out_dict['pred_slots'] = out_dict.pop('slots')
            out_dict['gt_slots'] = gt_slots
        else:
            pred_slots = self.rollout(past_slots, self.rollout_len, decode=False)
            out_dict = {'gt_slots': gt_slots, 'pred_slots': pred_slots}
        return out_dict

    def calc_train_loss(self, data_dict, out_dict):
        """slotformer.py:284-328."""
        loss_dict = {}
        gt_slots = out_dict['gt_slots']
        pred_slots = out_dict['pred_slots']
        slots_loss = F.mse_loss(pred_slots, gt_slots, reduction='none')
        if not self.training:
            for step in range(min(6, gt_slots.shape[1])):
                loss_dict[f'slot_recon_loss_{step+1}'] = slots_loss[:, step].mean()
        if self.loss_decay_factor < 1.:
            w = self.loss_decay_factor**torch.arange(gt_slots.shape[1])
            w = w.type_as(slots_loss)
            w = w / w.sum() * gt_slots.shape[1]
            slots_loss = slots_loss * w[None, :, None, None]
        vid_len = data_dict.get('vid_len', None)
        trunc_loss = False
        if (vid_len is not None) and (vid_len < (self.history_len + self.rollout_len)).any():
            trunc_loss = True
            valid_mask = torch.arange(gt_slots.shape[1]).to(gt_slots.device) + self.history_len
            valid_mask = valid_mask[None] < vid_len[:, None]
            valid_mask = valid_mask.flatten(0, 1)
            slots_loss = slots_loss.flatten(0, 1)[valid_mask]
        loss_dict['slot_recon_loss'] = slots_loss.mean()
        if self.use_img_recon_loss:
            recon_combined = out_dict['recon_combined']
            gt_img = data_dict['img'][:, self.history_len:]
            imgs_loss = F.mse_loss(recon_combined, gt_img, reduction='none')
            if trunc_loss:
                imgs_loss = imgs_loss.flatten(0, 1)[valid_mask]
            loss_dict['img_recon_loss'] = imgs_loss.mean()
        return loss_dict

    @property
    def dtype(self):
        return self.rollouter.dtype

    @property
    def device(self):
        return self.rollouter.device

    def train(self, mode=True):
        super().train(mode)
        self.decoder.eval()
        self.decoder_pos_embedding.eval()
        return self
