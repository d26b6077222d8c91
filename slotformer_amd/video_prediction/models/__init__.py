"""build_model for the rollout models (reference: slotformer/video_prediction/models/__init__.py:6-36)."""
from .slotformer import SlotFormer, SlotRollouter, get_sin_pos_enc, build_pos_enc
from .single_step_slotformer import SingleStepSlotFormer, SingleStepSlotRollouter
from .steve_slotformer import STEVESlotFormer


def build_model(params):
    kw = dict(
        resolution=params.resolution,
        clip_len=params.input_frames,
        slot_dict=params.slot_dict,
        dec_dict=params.dec_dict,
        rollout_dict=params.rollout_dict,
        loss_dict=params.loss_dict,
    )
    if params.model == 'SlotFormer':
        return SlotFormer(**kw)
    elif params.model == 'SingleStepSlotFormer':
        return SingleStepSlotFormer(**kw)
    elif params.model == 'STEVESlotFormer':
        return STEVESlotFormer(
            resolution=params.resolution,
            clip_len=params.input_frames,
            slot_dict=params.slot_dict,
            dvae_dict=params.dvae_dict,
            dec_dict=params.dec_dict,
            rollout_dict=params.rollout_dict,
            loss_dict=params.loss_dict,
        )
    else:
        raise NotImplementedError(f'{params.model} is not implemented.')
