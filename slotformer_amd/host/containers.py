"""Builders of the parameter containers of the slot models.

The classes in base_slots/models only own parameters; what a checkpoint cares about is the nesting of those containers
(it fixes the state-dict keys: `kernel_dist_layer.0.weight`, `encoder.2.0.bias`, `slot_attention.mlp.3.weight`, ...).
These helpers build the nestings from small shape descriptions."""
from torch import nn

from ..nerv_compat import conv_norm_act, deconv_norm_act, deconv_out_shape


def dense_stack(sizes, norm_after_first=False, norm_first=False):
    """nn.Sequential of Linear layers over `sizes`, ReLU between them.
    norm_first: a LayerNorm(sizes[0]) in front (indices shift by one); norm_after_first: LayerNorm after the first
    Linear (the `kernel_dist_layer` / `prior_slot_layer` shape: Linear, LayerNorm, ReLU, Linear)."""
    mods = [nn.LayerNorm(sizes[0])] if norm_first else []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i == 0 and norm_after_first:
            mods.append(nn.LayerNorm(sizes[1]))
        if i < len(sizes) - 2:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def conv_stack(channels, ks, norm, first_stride):
    """The SAVi CNN encoder: conv(+norm)+ReLU blocks, no activation after the last one; only the first conv may stride."""
    n = len(channels) - 1
    return nn.Sequential(*[
        conv_norm_act(channels[i], channels[i + 1], kernel_size=ks, stride=first_stride if i == 0 else 1, norm=norm,
                      act='relu' if i < n - 1 else '') for i in range(n)
    ])


def deconv_stack(channels, ks, norm, start_res, target_res, out_channels=4):
    """The spatial-broadcast decoder: stride-2 transposed convs until `target_res` is reached, stride 1 afterwards, then a
    1x1 conv to `out_channels`.  Returns (nn.Sequential, reached resolution)."""
    mods, size, stride = [], start_res, 2
    for i in range(len(channels) - 1):
        if size == target_res:
            stride = 1
        mods.append(deconv_norm_act(channels[i], channels[i + 1], kernel_size=ks, stride=stride, norm=norm, act='relu'))
        size = deconv_out_shape(size, stride, ks // 2, ks, stride - 1)
    mods.append(nn.Conv2d(channels[-1], out_channels, kernel_size=1, stride=1, padding=0))
    return nn.Sequential(*mods), size
