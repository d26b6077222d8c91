"""slotformer_amd: MI355X-native engine for the SlotFormer hot path.

SAVi/STEVE slot extraction (CNN encoder -> iterative Slot Attention + GRU update) and the
SlotFormer autoregressive Transformer rollout, as hand-written HIP kernels (gfx950) behind a
C ABI (include/slotformer_hip.h), exposed through the reference's own model-construction API:

    from slotformer_amd.base_slots import build_model        # StoSAVi / STEVE
    from slotformer_amd.video_prediction import build_model  # SlotFormer / SingleStepSlotFormer
"""
__version__ = '0.1.0'
