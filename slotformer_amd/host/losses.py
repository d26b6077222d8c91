"""Loss terms of the reference models (`calc_train_loss` / eval hooks) as torch ops on device tensors.  They are plain
differentiable reductions: in training (row N1) autograd carries their gradient into the HIP-backed nodes of train.py."""
import torch


def slot_rollout_losses(pred, gt, first_pred_frame, report_steps=False, decay=1.0, vid_len=None):
    """Squared-error terms of a slot rollout (slotformer.py:284-311).

    pred, gt [B,T,N,C].  Returns (dict of scalars, keep) where keep is the flattened [B*T] validity mask that was
    applied because some video is shorter than the rollout (None when nothing was truncated):
      * 'slot_recon_loss_k' for the first six steps when `report_steps` (the reference adds them in eval mode),
      * 'slot_recon_loss': mean over valid frames of the per-frame error, each frame t weighted by
        decay**t / sum(decay**t) * T when decay < 1.
    """
    err = (pred - gt)**2
    B, T = err.shape[:2]
    out = {}
    if report_steps:
        for i in range(min(6, T)):
            out[f'slot_recon_loss_{i + 1}'] = err[:, i].mean()
    if decay < 1.:
        w = (decay**torch.arange(T)).to(err.dtype).to(err.device)
        w = w / w.sum() * T
        err = err * w.view(1, T, 1, 1)
    keep = None
    if vid_len is not None and bool((vid_len < first_pred_frame + T).any()):
        frame_no = torch.arange(T, device=err.device) + first_pred_frame
        keep = (frame_no.unsqueeze(0) < vid_len.unsqueeze(1)).reshape(B * T)
        err = err.reshape(B * T, *err.shape[2:])[keep]
    out['slot_recon_loss'] = err.mean()
    return out, keep


def image_recon_loss(recon, target, keep=None):
    """Mean squared image error over the kept frames (slotformer.py:313-326, savi.py:527-538)."""
    err = (recon - target)**2
    if keep is not None:
        err = err.reshape(keep.shape[0], *err.shape[2:])[keep]
    return err.mean()


def kernel_kld(dist, slot_size, prior_log_var):
    """KL( N(mu, exp(log_var)) || N(mu, exp(prior_log_var)) ) summed over channels, averaged over slots / frames -- the
    regulariser of the stochastic SAVi kernels (savi.py:337-353).  dist [..., 2*slot_size] = (mu | log_var); the prior
    shares the posterior mean, so only the variances enter."""
    assert dist.shape[-1] == 2 * slot_size
    log_var = dist[..., slot_size:]
    prior = torch.full_like(log_var, prior_log_var)
    sigma, sigma_p = torch.exp(0.5 * log_var), torch.exp(0.5 * prior)
    kld = torch.log(sigma_p / sigma) + torch.exp(log_var) / (2. * torch.exp(prior)) - 0.5
    return kld.sum(-1).mean()
