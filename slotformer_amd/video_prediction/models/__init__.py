"""Rollout models and their factory (API of slotformer/video_prediction/models/__init__.py)."""
from ...host.registry import ModelTable, ROLLOUT_MODEL_ARGS
from .slotformer import SlotFormer, SlotRollouter, get_sin_pos_enc, build_pos_enc
from .single_step_slotformer import SingleStepSlotFormer, SingleStepSlotRollouter
from .steve_slotformer import STEVESlotFormer

_TABLE = (ModelTable()
          .add('SlotFormer', SlotFormer, **ROLLOUT_MODEL_ARGS)
          .add('SingleStepSlotFormer', SingleStepSlotFormer, **ROLLOUT_MODEL_ARGS)
          .add('STEVESlotFormer', STEVESlotFormer, dvae_dict='dvae_dict', **ROLLOUT_MODEL_ARGS))


def build_model(params):
    """params.model in {'SlotFormer', 'SingleStepSlotFormer', 'STEVESlotFormer'} -> the model (NotImplementedError else)."""
    return _TABLE.build(params)
