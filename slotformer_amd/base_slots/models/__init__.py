"""build_model for the slot-extraction models (reference: slotformer/base_slots/models/__init__.py:9-34)."""
from .savi import StoSAVi, SlotAttention
from .steve import STEVE, SlotAttentionWMask
from .utils import to_rgb_from_tensor, assert_shape, SoftPositionEmbed, build_grid


def build_model(params):
    if params.model == 'StoSAVi':
        return StoSAVi(
            resolution=params.resolution,
            clip_len=params.input_frames,
            slot_dict=params.slot_dict,
            enc_dict=params.enc_dict,
            dec_dict=params.dec_dict,
            pred_dict=params.pred_dict,
            loss_dict=params.loss_dict,
        )
    elif params.model == 'STEVE':
        return STEVE(
            resolution=params.resolution,
            clip_len=params.input_frames,
            slot_dict=params.slot_dict,
            dvae_dict=params.dvae_dict,
            enc_dict=params.enc_dict,
            dec_dict=params.dec_dict,
            pred_dict=params.pred_dict,
            loss_dict=params.loss_dict,
        )
    elif params.model == 'dVAE':
        raise NotImplementedError('dVAE (image tokenizer) is outside the slot-extraction/rollout hot path')
    else:
        raise NotImplementedError(f'{params.model} is not implemented.')
