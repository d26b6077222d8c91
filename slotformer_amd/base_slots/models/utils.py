"""Small helpers with the names of slotformer/base_slots/models/utils.py (the reference's scripts import them)."""
import torch
from torch import nn


def torch_cat(tensor_list, dim):
    """torch.cat along `dim`, or along dim 0 when the tensors have too few dimensions for it."""
    return torch.cat(tensor_list, dim=dim) if tensor_list[0].dim() > dim else torch.cat(tensor_list)


def assert_shape(actual, expected, message=''):
    if list(actual) != list(expected):
        raise AssertionError(f'Expected shape: {expected} but passed shape: {actual}. {message}')


def build_grid(resolution):
    """[1, H, W, 4]: (y, x, 1 - y, 1 - x) with y, x = linspace(0, 1) over rows / columns (utils.py:37-44)."""
    H, W = resolution
    y = torch.linspace(0.0, 1.0, steps=H).view(H, 1).expand(H, W)
    x = torch.linspace(0.0, 1.0, steps=W).view(1, W).expand(H, W)
    yx = torch.stack((y, x), dim=-1)
    return torch.cat((yx, 1.0 - yx), dim=-1).unsqueeze(0).contiguous()


def to_rgb_from_tensor(x):
    """[-1, 1] image tensor -> [0, 1]."""
    return (0.5 * x + 0.5).clamp(0, 1)


class SoftPositionEmbed(nn.Module):
    """Owner of the soft position embedding's parameters (utils.py:52-63): `dense.weight [C,4]`, `dense.bias [C]` and the
    buffer `grid [1,H,W,4]` (the grid IS part of the reference's state dict).  The engine folds them into a [H*W, C] table
    once per weight version (`sf_pos_embed_table_f32`) and adds the table in the epilogue of the last encoder conv."""

    def __init__(self, hidden_size, resolution):
        super().__init__()
        self.dense = nn.Linear(in_features=4, out_features=hidden_size)
        self.register_buffer('grid', build_grid(resolution))
