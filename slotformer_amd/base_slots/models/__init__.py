"""build_model for the slot-extraction models (reference: slotformer/base_slots/models/__init__.py:9-34)."""
from .savi import StoSAVi, SlotAttention
from .dVAE import dVAE
from .steve import STEVE, SlotAttentionWMask
from .steve_transformer import STEVETransformerDecoder
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
        return dVAE(vocab_size=params.vocab_size, img_channels=3)
    else:
        raise NotImplementedError(f'{params.model} is not implemented.')
