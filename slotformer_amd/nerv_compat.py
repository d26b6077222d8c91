"""Minimal stand-ins for the `nerv` symbols the reference models are written against.

Reconstructed from the reference's call sites only -- the nerv package
(Wuziyi616/nerv v0.1.0) is third-party and not part of the reference tree.
"""
import torch
import torch.nn as nn


class BaseModel(nn.Module):
    """forward(data_dict) -> out_dict; calc_train_loss(data_dict, out_dict) -> {name: loss}."""

    def forward(self, data_dict):
        raise NotImplementedError

    def calc_train_loss(self, data_dict, out_dict):
        raise NotImplementedError

    @torch.no_grad()
    def calc_eval_loss(self, data_dict, out_dict):
        return self.calc_train_loss(data_dict, out_dict)

    def loss_function(self, data_dict):
        """forward + the loss dict of the current mode: what nerv's training loop calls on the model before it sums the
        terms with the `<name>_w` weights of the params file (reconstructed from nerv v0.1.0's public behaviour -- the
        package is not in the reference tree, so this wrapper is parity-unpinned)."""
        out_dict = self.forward(data_dict)
        return self.calc_train_loss(data_dict, out_dict) if self.training else self.calc_eval_loss(data_dict, out_dict)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseParams:
    """Attribute container used by the reference's *_params.py files."""

    def get(self, name, default=None):
        return getattr(self, name, default)


def conv_norm_act(in_channels, out_channels, kernel_size, stride=1, norm='', act='relu'):
    """nerv.models.conv_norm_act as the reference uses it (savi.py:231-239): 'same' padding,
    Sequential(conv, norm, act) nesting (state-dict keys `encoder.{i}.0.*`).  Only norm=''."""
    if norm:
        raise NotImplementedError(f"enc_norm={norm!r}: only norm='' is supported (all reference configs)")
    return nn.Sequential(
        nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2),
        nn.Identity(), nn.ReLU() if act == 'relu' else nn.Identity())


def deconv_norm_act(in_channels, out_channels, kernel_size, stride=1, norm='', act='relu'):
    """nerv.models.deconv_norm_act (savi.py:269-275)."""
    if norm:
        raise NotImplementedError(f"dec_norm={norm!r}: only norm='' is supported")
    return nn.Sequential(
        nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2,
                           output_padding=stride - 1), nn.Identity(),
        nn.ReLU() if act == 'relu' else nn.Identity())


def deconv_out_shape(in_size, stride, padding, kernel_size, out_padding):
    return (in_size - 1) * stride - 2 * padding + kernel_size + out_padding
