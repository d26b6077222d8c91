"""Mask post-processing used by the video-prediction evaluation (reference: vp_utils.py:20-41)."""
import torch

FG_THRE = 0.5


def postproc_mask(batch_masks):
    """[B,T,N,1,H,W] soft masks -> int64 [B,T,H,W]: the slot whose peak score is smallest is the
    background; pixels whose best score is below FG_THRE are assigned to it, everything else is an
    argmax over slots."""
    m = batch_masks.clone()
    B, T, N, _, H, W = m.shape
    m = m.reshape(B * T, N, H * W)
    bg_idx = m.max(-1)[0].argmin(-1)
    weak = m.max(1)[0] < FG_THRE
    is_bg = torch.zeros(B * T, N, dtype=torch.bool, device=m.device)
    is_bg[torch.arange(B * T, device=m.device), bg_idx] = True
    m[is_bg.unsqueeze(-1) & weak.unsqueeze(1)] = 1.
    return m.argmax(1).reshape(B, T, H, W)
