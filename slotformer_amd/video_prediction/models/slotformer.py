"""SlotFormer (slots in -> future slots out) on the MI355X engine.

Same public surface as the reference's slotformer/video_prediction/models/slotformer.py -- class and attribute names,
constructor keywords, output dictionaries, state-dict keys (`rollouter.*`, frozen `decoder.*`) -- because the reference's
scripts poke these from outside (`rollout_len`, `use_img_recon_loss`, `loss_decay_factor`, `testing`).  What differs is
everything underneath: the modules only own parameters; the autoregressive loop is one call into libslotformer_hip
(`sf_rollout_f32`), which keeps every frame's slots in a single [B, T, N, C] device buffer, reads each step's window as
rows of that buffer and writes the prediction in place.
"""
import torch
from torch import nn

from ...nerv_compat import BaseModel
from ... import engine, train
from ...base_slots.models import StoSAVi
from ...host import frozen, losses

ROLLOUTER_DEFAULTS = dict(num_slots=7, slot_size=128, history_len=6, t_pe='sin', slots_pe='', d_model=128, num_layers=4,
                          num_heads=8, ffn_dim=512, norm_first=True)
DECODER_DEFAULTS = dict(dec_channels=(128, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5, dec_norm='', dec_ckp_path='')


def get_sin_pos_enc(seq_len, d_model):
    """Sinusoid table [1, seq_len, d_model] whose LAST row is position 0 (the newest frame), first half sines, second half
    cosines -- the convention of slotformer.py:10-16, which a checkpoint's `enc_t_pe` buffer was trained with."""
    exponent = torch.arange(0.0, d_model, 2.0) / d_model
    inv_freq = 1. / (10000**exponent)
    age = torch.arange(seq_len - 1, -1, -1).type_as(inv_freq)   # how many frames back each row is
    phase = torch.outer(age, inv_freq)
    return torch.cat((phase.sin(), phase.cos()), dim=-1)[None]


def build_pos_enc(pos_enc, input_len, d_model):
    """'' -> None; 'learnable' -> zero-initialised trainable table; '...sin...' -> fixed sinusoid table.  Always an
    nn.Parameter so that it is part of the state dict, as in the reference (slotformer.py:19-32)."""
    if not pos_enc:
        return None
    if pos_enc == 'learnable':
        return nn.Parameter(torch.zeros(1, input_len, d_model))
    if 'sin' not in pos_enc:
        raise NotImplementedError(f'unsupported pos enc {pos_enc}')
    return nn.Parameter(get_sin_pos_enc(input_len, d_model), requires_grad=False)


class Rollouter(nn.Module):
    """Interface of the reference's rollouters; both hooks are no-ops for the Transformer rollouter."""

    def burnin(self, x):
        return None

    def reset(self):
        return None


class SlotRollouter(Rollouter):
    """Parameters of the Transformer rollouter (slotformer.py:48-134): in/out projections, a pre-LN
    nn.TransformerEncoder and the temporal / slot position tables.  `forward` hands the whole rollout to the engine."""

    def __init__(self, num_slots, slot_size, history_len, t_pe='sin', slots_pe='', d_model=128, num_layers=4, num_heads=8,
                 ffn_dim=512, norm_first=True):
        super().__init__()
        self.num_slots, self.history_len = num_slots, history_len
        self.in_proj = nn.Linear(slot_size, d_model)
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=num_heads, dim_feedforward=ffn_dim, norm_first=norm_first,
                                           batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(encoder_layer=layer, num_layers=num_layers, enable_nested_tensor=False)
        self.enc_t_pe = build_pos_enc(t_pe, history_len, d_model)
        self.enc_slots_pe = build_pos_enc(slots_pe, num_slots, d_model)
        self.out_proj = nn.Linear(d_model, slot_size)

    def _n_in(self):
        return self.history_len

    def forward(self, x, pred_len):
        """x [B, history_len, N, C] burn-in slots -> the pred_len predicted frames [B, pred_len, N, C]."""
        assert x.shape[1] == self.history_len, 'wrong burn-in steps'
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return train.rollout_with_grad(self, x, pred_len)   # one autograd node around the HIP forward / backward
        n_in = x.shape[1]
        frames = torch.empty(x.shape[0], n_in + pred_len, *x.shape[2:], device=x.device, dtype=torch.float32)
        frames[:, :n_in].copy_(x)
        engine.rollout(self, frames, n_in, pred_len)
        return frames[:, n_in:]

    @property
    def dtype(self):
        return self.in_proj.weight.dtype

    @property
    def device(self):
        return self.in_proj.weight.device


class SlotFormer(BaseModel):
    """Autoregressive dynamics model over slots (slotformer.py:137-343) with the frozen SAVi decoder for visualisation /
    image losses."""

    def __init__(self, resolution, clip_len, slot_dict=None, dec_dict=None, rollout_dict=None, loss_dict=None, eps=1e-6):
        super().__init__()
        self.resolution, self.clip_len, self.eps = resolution, clip_len, eps
        self.slot_dict = slot_dict if slot_dict is not None else dict(num_slots=7, slot_size=128)
        self.dec_dict = dec_dict if dec_dict is not None else dict(DECODER_DEFAULTS)
        self.rollout_dict = rollout_dict if rollout_dict is not None else dict(ROLLOUTER_DEFAULTS)
        self.loss_dict = loss_dict if loss_dict is not None else dict(rollout_len=6, use_img_recon_loss=False)
        # construction order fixes the state-dict order: decoder (+ its position embedding), then the rollouter
        self._build_slot_attention()
        self._build_decoder()
        self._build_rollouter()
        self._build_loss()
        self.testing = False          # kept for the scripts that set it
        self.loss_decay_factor = 1.   # per-step loss weight decay, set by the training method

    # ---- construction -------------------------------------------------------------------------------------------
    def _build_slot_attention(self):
        self.num_slots, self.slot_size = self.slot_dict['num_slots'], self.slot_dict['slot_size']

    def _build_decoder(self):
        """The SAVi spatial-broadcast decoder, filled from the `decoder.*` / `decoder_pos_embedding.*` entries of the
        checkpoint at dec_dict['dec_ckp_path'] and frozen."""
        StoSAVi._build_decoder(self)
        sd = frozen.checkpoint_state(self.dec_dict['dec_ckp_path'], 'decoder')
        frozen.load_prefixed(self.decoder, sd, 'decoder.')
        frozen.load_prefixed(self.decoder_pos_embedding, sd, 'decoder_pos_embedding.')
        frozen.freeze(self.decoder, self.decoder_pos_embedding)

    def _build_rollouter(self):
        self.history_len = self.rollout_dict['history_len']
        self.rollouter = SlotRollouter(**self.rollout_dict)

    def _build_loss(self):
        self.rollout_len = self.loss_dict['rollout_len']
        self.use_img_recon_loss = self.loss_dict['use_img_recon_loss']

    # ---- inference ----------------------------------------------------------------------------------------------
    def decode(self, slots):
        return StoSAVi.decode(self, slots)

    def rollout(self, past_slots, pred_len, decode=False, with_gt=True):
        """Predict pred_len frames from the last history_len frames of past_slots.  decode=False: the predicted slots.
        decode=True: a dict with the decoded frames of (past + predicted) slots, or of the predicted ones only when
        with_gt is False (slotformer.py:236-261)."""
        pred = self.rollouter(past_slots[:, -self.history_len:], pred_len)
        if not decode:
            return pred
        shown = torch.cat((past_slots, pred), dim=1) if with_gt else pred
        B, T = shown.shape[:2]
        recon, per_slot, masks, _ = self.decode(shown.flatten(0, 1))
        out = {name: t.unflatten(0, (B, T)) for name, t in (('recon_combined', recon), ('recons', per_slot), ('masks', masks))}
        out['slots'] = shown
        return out

    def forward(self, data_dict):
        """{'slots': [B, history_len + rollout_len, N, C]} -> {'pred_slots', 'gt_slots'} (+ decoded frames when the image
        loss is on)."""
        slots = data_dict['slots']
        assert self.rollout_len + self.history_len == slots.shape[1], f'wrong SlotFormer training length {slots.shape[1]}'
        burn_in, target = slots[:, :self.history_len], slots[:, self.history_len:]
        if not self.use_img_recon_loss:
            return {'gt_slots': target, 'pred_slots': self.rollout(burn_in, self.rollout_len, decode=False)}
        out = self.rollout(burn_in, self.rollout_len, decode=True, with_gt=False)
        out['pred_slots'] = out.pop('slots')
        out['gt_slots'] = target
        return out

    def calc_train_loss(self, data_dict, out_dict):
        """Loss values of slotformer.py:284-328 (see host/losses.py for the weighting / truncation rules)."""
        terms, keep = losses.slot_rollout_losses(out_dict['pred_slots'], out_dict['gt_slots'], self.history_len,
                                                 report_steps=not self.training, decay=self.loss_decay_factor,
                                                 vid_len=data_dict.get('vid_len', None))
        if self.use_img_recon_loss:
            terms['img_recon_loss'] = losses.image_recon_loss(out_dict['recon_combined'],
                                                              data_dict['img'][:, self.history_len:], keep)
        return terms

    @property
    def dtype(self):
        return self.rollouter.dtype

    @property
    def device(self):
        return self.rollouter.device

    def train(self, mode=True):
        super().train(mode)
        frozen.freeze(self.decoder, self.decoder_pos_embedding)   # the borrowed decoder never leaves eval mode
        return self
