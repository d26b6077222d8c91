"""Slot-extraction models and their factory (API of slotformer/base_slots/models/__init__.py)."""
from ...host.registry import ModelTable, SLOT_MODEL_ARGS
from .savi import StoSAVi, SlotAttention
from .dVAE import dVAE
from .steve import STEVE, SlotAttentionWMask
from .steve_transformer import STEVETransformerDecoder
from .utils import to_rgb_from_tensor, assert_shape, SoftPositionEmbed, build_grid




def _dvae_from_params(vocab_size):
    """`params.vocab_size` is the only attribute the reference's factory reads for the dVAE."""
    return dVAE(vocab_size=vocab_size, img_channels=3)


_TABLE = (ModelTable()
          .add('StoSAVi', StoSAVi, **SLOT_MODEL_ARGS)
          .add('STEVE', STEVE, dvae_dict='dvae_dict', **SLOT_MODEL_ARGS)
          .add('dVAE', _dvae_from_params, vocab_size='vocab_size'))


def build_model(params):
    """params.model in {'StoSAVi', 'STEVE', 'dVAE'} -> the model (NotImplementedError otherwise)."""
    return _TABLE.build(params)
