"""Shared definitions for golden-vector cases (used by tools/gen_golden.py and tests/).

Weights and inputs are regenerated from seeds (legacy ``RandomState`` -- bit-stable
across NumPy versions) so the committed fixtures only hold *outputs* of the
reference classes plus the (key, shape) list of the reference state dict.
"""
import os
import zlib

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# keys whose value is a deterministic closed form (kept as constructed, and
# compared against the reference's value stored in the fixture)
CLOSED_FORM_SUFFIXES = ('.grid', 'enc_t_pe', 'self_attn_mask')
CLOSED_FORM_NOSTORE = ('self_attn_mask', )  # boolean triu masks: taken from the module under test, never stored


def seeded_tensor(key, shape, seed):
    rs = np.random.RandomState((zlib.crc32(key.encode()) ^ seed) & 0x7fffffff)
    shape = tuple(shape)
    if key.endswith('init_latents'):
        a = rs.standard_normal(shape)
    elif key.endswith(('tok_emb.weight', 'pos_emb.pe')):  # embeddings: O(1) entries
        a = 0.5 * rs.standard_normal(shape)
    elif 'bias' in key.rsplit('.', 1)[-1]:
        a = 0.1 * (rs.rand(*shape) * 2 - 1)
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        a = (rs.rand(*shape) * 2 - 1) * (1.5 / np.sqrt(fan_in))
    else:  # 1-D "weight" = LayerNorm gamma
        a = 1.0 + 0.1 * rs.standard_normal(shape)
    return torch.from_numpy(a.astype(np.float32))


def seeded_state_dict(shapes, seed, keep=None):
    """shapes: list of (key, shape).  keep: dict of closed-form tensors to reuse."""
    sd = {}
    for key, shape in shapes:
        if key.endswith(CLOSED_FORM_NOSTORE) and (keep is None or key not in keep):
            sd[key] = torch.triu(torch.ones(tuple(shape), dtype=torch.bool), diagonal=1)  # steve_transformer.py:163-165
        elif key.endswith(CLOSED_FORM_SUFFIXES):
            assert keep is not None and key in keep, key
            sd[key] = keep[key].clone()
        else:
            sd[key] = seeded_tensor(key, shape, seed)
    return sd


def seeded_img(B, T, res, seed=1234):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.rand(B, T, 3, res, res) * 2 - 1).astype(np.float32))


def seeded_normal(shape, seed):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


from slotformer_amd.configs import *  # noqa: F401,F403,E402  (C1..C5 configuration dicts, savi_cfg, rollout_cfg, ParamsView)
from slotformer_amd.configs import savi_cfg, rollout_cfg, ParamsView, steve_tokens_cfg, steve_slotformer_cfg  # noqa: F401,E402


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def shapes_from_golden(g):
    keys = [str(k) for k in g['sd_keys']]
    shapes = [tuple(int(v) for v in s.split('x')) if s else () for s in (str(x) for x in g['sd_shapes'])]
    return list(zip(keys, shapes))
