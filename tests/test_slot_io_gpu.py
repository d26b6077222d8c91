"""Row N3 on the GPU box: engine run -> the reference's slot files -> read back -> rollout -> per-sample files, equal to the
in-memory path (extract_slots.py:57-93, datasets/clevrer.py:323-335, rollout_clevrer_slots.py:20-65,
extract_phyre_slots.py:45-76 incl. its restart rule)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
from test_engine_gpu import build

pytestmark = pytest.mark.gpu


@torch.no_grad()
def test_extract_dump_load_rollout_roundtrip(dev, tmp_path):
    from slotformer_amd import harness, slot_io
    g = gu.load_golden('savi_c1')
    savi, _ = build(gu.C1_SAVI, g, 101, dev)
    savi.testing = True
    sf, _ = build(gu.C1_ROLL, gu.load_golden('roll_c1'), 201, dev, vp=True)
    V, T = 4, 8
    vids = gu.seeded_img(V, T, 64, seed=77)
    files = [f'/data/CLEVRER/videos/train/video_{i:05d}.mp4' for i in range(V)]
    slots = harness.extract_video_slots(savi, vids, batch_size=2)          # H1: [V,T,N,D] on the host
    path = str(tmp_path / 'slots' / 'clevrer_slots.pkl')
    slot_io.dump_slots(path, train=slot_io.slots_to_dict(files[:3], slots[:3]), val=slot_io.slots_to_dict(files[3:], slots[3:]))
    back = slot_io.load_slots(path)
    assert set(back) == {'train', 'val'} and set(back['train']) == {os.path.basename(f) for f in files[:3]}
    for i, f in enumerate(files):
        split = 'train' if i < 3 else 'val'
        assert np.array_equal(back[split][os.path.basename(f)], slots[i].numpy())
    # dataset side: strided clips read back by basename (datasets/clevrer.py:323-335) == slicing the in-memory slots
    clip = slot_io.read_clip(back['train'], files[1], start_idx=1, n_sample_frames=3, frame_offset=2)
    assert np.array_equal(clip, slots[1][1:6:2].numpy())
    # rollout from the FILE contents == rollout from the in-memory slots (the file holds exact float32)
    hist = sf.history_len
    ori_file = torch.from_numpy(np.stack([back['train'][os.path.basename(f)] for f in files[:3]]))[:, :hist]
    out_file = harness.rollout_video_slots(sf, ori_file, frame_offset=1, obs_frames=hist, target_len=hist + 5)
    out_mem = harness.rollout_video_slots(sf, slots[:3, :hist], frame_offset=1, obs_frames=hist, target_len=hist + 5)
    assert torch.equal(out_file, out_mem) and out_file.shape == (3, hist + 5, 6, 128)
    # PHYRE layout: one .npy per sample truncated to the video length, restartable
    root = str(tmp_path / 'phyre')
    for i in range(3):
        slot_io.save_phyre_slots(root, 100 + i, out_file[i].cpu().numpy(), vid_len=hist + 3)
    assert np.array_equal(np.load(slot_io.phyre_path(root, 101)), out_file[1, :hist + 3].cpu().numpy())
    assert slot_io.phyre_resume_index(root, 100, 110) == 102            # stop at the first gap, redo the one before it
    os.remove(slot_io.phyre_path(root, 101))
    assert slot_io.phyre_resume_index(root, 100, 110) == 100
