"""Mask post-processing used by the video-prediction evaluation (reference: vp_utils.py:20-41)."""
import torch

FG_THRE = 0.5


def postproc_mask(batch_masks):
    """[B,T,N,1,H,W] soft masks -> int64 [B,T,H,W]: the slot whose peak score is smallest is the
    background; pixels whose best score is below FG_THRE are assigned to it, everything else is an
    argmax over slots."""
    if batch_masks.is_cuda and batch_masks.dtype == torch.float32 and batch_masks.shape[2] <= 255:
        # device tensors: two launches of the HIP library (per-(frame, slot) maxima, then the rule per pixel); the same comparisons
        import ctypes as C  # noqa: F401
        from .. import _lib
        B, T, N, _, H, W = batch_masks.shape
        mk = batch_masks.detach().contiguous()
        out = torch.empty(B, T, H, W, dtype=torch.int64, device=mk.device)
        scratch = torch.empty(B * T * N, dtype=torch.int32, device=mk.device)
        _lib.check(_lib.lib().sf_postproc_mask_f32(mk.data_ptr(), out.data_ptr(), None, float(FG_THRE), scratch.data_ptr(), B * T, N, H * W,
                                                   torch.cuda.current_stream(mk.device).cuda_stream))
        return out
    m = batch_masks.clone()
    B, T, N, _, H, W = m.shape
    m = m.reshape(B * T, N, H * W)
    bg_idx = m.max(-1)[0].argmin(-1)
    weak = m.max(1)[0] < FG_THRE
    is_bg = torch.zeros(B * T, N, dtype=torch.bool, device=m.device)
    is_bg[torch.arange(B * T, device=m.device), bg_idx] = True
    m[is_bg.unsqueeze(-1) & weak.unsqueeze(1)] = 1.
    return m.argmax(1).reshape(B, T, H, W)
