"""Helpers mirroring slotformer/base_slots/models/utils.py (reference file:line in docstrings)."""
import torch
import torch.nn as nn


def torch_cat(tensor_list, dim):
    if len(tensor_list[0].shape) <= dim:
        return torch.cat(tensor_list)
    return torch.cat(tensor_list, dim=dim)


def assert_shape(actual, expected, message=""):
    assert list(actual) == list(expected), \
        f"Expected shape: {expected} but passed shape: {actual}. {message}"


def build_grid(resolution):
    """[1, H, W, 4] grid of (y, x, 1-y, 1-x) in [0, 1].  utils.py:37-44."""
    axes = [torch.linspace(0.0, 1.0, steps=r) for r in resolution]
    yx = torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1)
    yx = yx.reshape(resolution[0], resolution[1], -1).unsqueeze(0)
    return torch.cat([yx, 1.0 - yx], dim=-1)


def to_rgb_from_tensor(x):
    return (x * 0.5 + 0.5).clamp(0, 1)


class SoftPositionEmbed(nn.Module):
    """Parameter container for the soft position embedding (utils.py:52-63).

    State-dict keys: `dense.weight [C,4]`, `dense.bias [C]`, buffer `grid [1,H,W,4]`.
    The engine turns it into a [H*W, C] table once (sf_pos_embed_table_f32) and adds it in the
    epilogue of the last encoder conv.
    """

    def __init__(self, hidden_size, resolution):
        super().__init__()
        self.dense = nn.Linear(in_features=4, out_features=hidden_size)
        self.register_buffer('grid', build_grid(resolution))
