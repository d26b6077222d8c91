"""On-disk slot formats the reference's offline pipeline exchanges (row N3 of SURVEY.md 8f).

* CLEVRER / OBJ3D / Physion: one pickle per dataset, ``{split: {video_basename: float32 [T,N,C]}}``
  (extract_slots.py:57-76, written with nerv ``dump_obj`` = pickle for ``.pkl``); read back by key
  ``os.path.basename(video_path)`` (datasets/clevrer.py:323-335).
* PHYRE: one ``{data_idx:06d}.npy`` per sample holding ``slots[:vid_len]`` (extract_phyre_slots.py:69-76);
  extraction is restartable -- files already present are skipped, the newest one is redone in case it
  was truncated (extract_phyre_slots.py:45-53).
"""
import os
import pickle

import numpy as np


def slots_to_dict(files, slots):
    """files: list of video paths; slots: array-like [V,T,N,C] -> {basename: float32 [T,N,C]}."""
    assert len(files) == len(slots)
    return {os.path.basename(f): np.asarray(s, dtype=np.float32) for f, s in zip(files, slots)}


def dump_slots(path, **splits):
    """dump_slots('slots.pkl', train={...}, val={...}[, test={...}]) -- the reference's pickle layout."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'wb') as f:
        pickle.dump({k: v for k, v in splits.items()}, f)


def link_slots(save_path, weight_path, subset=None):
    """The soft link the reference leaves next to the weights a slot file was extracted with (extract_slots.py:86-93): `<weight dir>/slots.pkl`,
    or `<weight dir>/<subset>_slots.pkl` for the Physion subsets -- where the video-prediction configs look for the slots.  An existing link is
    replaced (the reference's `ln -s` fails silently there); returns the link's path."""
    name = 'slots.pkl' if subset is None else f'{subset}_slots.pkl'
    ln_path = os.path.join(os.path.dirname(os.path.abspath(weight_path)), name)
    if os.path.islink(ln_path):
        os.unlink(ln_path)
    os.symlink(os.path.abspath(save_path), ln_path)
    return ln_path


def load_slots(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def read_clip(video_slots, video_path, start_idx, n_sample_frames, frame_offset):
    """datasets/clevrer.py:323-335: strided clip [n_sample_frames,N,C] of one video's slots."""
    try:
        slots = video_slots[os.path.basename(video_path)]
    except KeyError:
        raise ValueError(f'no slots for {video_path}')
    return np.stack([slots[start_idx + n * frame_offset] for n in range(n_sample_frames)], 0).astype(np.float32)


def phyre_path(save_root, data_idx):
    return os.path.join(save_root, f'{int(data_idx):06d}.npy')


def save_phyre_slots(save_root, data_idx, slots, vid_len):
    """One file per sample, truncated to the real video length (extract_phyre_slots.py:69-76)."""
    os.makedirs(save_root, exist_ok=True)
    np.save(phyre_path(save_root, data_idx), np.asarray(slots, dtype=np.float32)[:int(vid_len)])


def phyre_resume_index(save_root, start, end):
    """First index in [start, end) that still has to be processed (extract_phyre_slots.py:45-53): scan forward, stop at
    the FIRST missing file and redo the one before it (it may have been truncated by the crash).  Files beyond a gap
    are not trusted -- every index from the returned one on is (re)written."""
    idx = start
    while idx < end and os.path.exists(phyre_path(save_root, idx)):
        idx += 1
    return max(idx - 1, start)
