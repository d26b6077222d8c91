"""Counterparts of the reference's offline drivers around the hot path.

The reference scripts mix dataset I/O, nn.DataParallel and pickling with a few lines of index
arithmetic; only that arithmetic and the model calls are reproduced here, operating on tensors.
"""
import torch

OBS_FRAMES = 128   # rollout_clevrer_slots.py:15
TARGET_LEN = 160   # rollout_clevrer_slots.py:16


@torch.no_grad()
def extract_video_slots(model, videos, batch_size=1):
    """H1 -- extract_slots.py:19-38.  videos: [V, T, 3, H, W] (tensor or list of [T,3,H,W]) ->
    float32 numpy-ready tensor [V, T, N, D] on the host.  `model` is a StoSAVi / STEVE in eval
    mode with testing=True; whole videos are encoded, `batch_size` videos per engine call."""
    model.eval()
    assert model.testing, 'set model.testing = True for slot extraction (extract_slots.py:127)'
    key = 'post_slots' if hasattr(model, 'kernel_dist_layer') else 'slots'
    out = []
    n = len(videos)
    for s in range(0, n, batch_size):
        clip = videos[s:s + batch_size]
        if not torch.is_tensor(clip):
            clip = torch.stack(list(clip), 0)
        res = model({'img': clip.float().to(model.device)})
        out.append(res[key].detach().cpu())
    return torch.cat(out, 0)


@torch.no_grad()
def rollout_video_slots(model, ori_slots, frame_offset, history_len=None, obs_frames=OBS_FRAMES,
                        target_len=TARGET_LEN):
    """H2 -- rollout_clevrer_slots.py:20-65 / rollout_physion_slots.py:22-63.

    ori_slots [B, obs_frames, N, C] -> [B, target_len, N, C]: for every offset `off` the strided
    sub-sequence start::frame_offset (start = obs - hist*offset + off) is rolled out for as many
    steps as it has frames beyond the burn-in, and the predictions are interleaved back.
    """
    model.eval()
    hist = model.history_len if history_len is None else history_len
    dev = model.device
    ori = ori_slots.float().to(dev)
    B, _, N, C = ori.shape
    full = torch.cat([ori, torch.zeros(B, target_len - obs_frames, N, C, device=dev)], 1)
    preds = []
    for off in range(frame_offset):
        start = obs_frames - hist * frame_offset + off
        in_slots = full[:, start::frame_offset].contiguous()
        model.rollout_len = in_slots.shape[1] - hist
        preds.append(model({'slots': in_slots})['pred_slots'])
    pred = torch.stack([preds[i % frame_offset][:, i // frame_offset] for i in range(target_len - obs_frames)], 1)
    out = torch.cat([full[:, :obs_frames], pred], 1)
    assert out.shape[1] == target_len
    return out


@torch.no_grad()
def encode_then_rollout(savi, slotformer, img0, vid_len, noise=None):
    """H3 -- test_phyre_planning.py:159-174: SAVi on the first frame(s), zero-pad to `vid_len`,
    SlotFormer forward, all on device.  img0 [B, T0, 3, H, W]."""
    data = {'img': img0.float().to(savi.device)}
    if noise is not None:
        data['noise'] = noise.to(savi.device)
    slot0 = savi(data)['post_slots']
    B, T0, N, C = slot0.shape
    slots = torch.zeros(B, vid_len, N, C, device=slot0.device)
    slots[:, :T0] = slot0
    slotformer.rollout_len = vid_len - slotformer.history_len
    return slotformer({'slots': slots})
