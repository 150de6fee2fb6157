"""Spectral normalisation of the reference's ``models/sn_utils.py`` on the HIP power-iteration op: ``max_singular_value``
(:11-25), ``SNLinear`` (:28-47), ``SNConv2d`` (:49-72).  Upstream only the unreachable ``Res_Discriminator`` uses them
(Discriminator_NET.py:386-455); here they are standalone layers plus the build's optional ``--sn_D`` wrap of the
multi-scale PatchGAN's convolutions (models/Discriminator_NET.py).  Parity pinned at the layer level against the real
reference classes (tests/golden/sn_layers.npz)."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..nn import Conv2d, _Frozen


def max_singular_value(W, u=None, Ip=1):
    if Ip != 1:
        raise NotImplementedError('Ip=1 only (the reference default)')
    return ops.sn_max_singular_value(W, u)


class _SNMixin(object):
    """W_bar = W / sigma(W); ``u`` (1 x rows) is overwritten by every training-mode forward (reference :38-42, :62-67).
    The reference re-registers ``u`` as a non-grad Parameter each time; here it is a buffer under the same state_dict key
    (not an optimizer parameter: Adam never sees it on either side -- its grad is None upstream)."""
    Ip = 1

    def _init_u(self, rows):
        self.register_buffer('u', torch.randn(1, rows))

    @property
    def W_bar(self):
        # frozen_params(): the discriminator pass inside loss_G -- no gradient may reach D's weights
        w = self.weight.detach() if _Frozen.on else self.weight
        sigma, _u = max_singular_value(w, self.u, self.Ip)
        if self.training:
            with torch.no_grad():
                self.u.copy_(_u)
        wb = ops.div_scalar(w, sigma)
        wb._him_wkey = id(self.weight)      # gradient routing keys (ops.SKIP_WGRAD / SKIP_DGRAD) name the PARAMETER
        return wb


class SNConv2d(_SNMixin, Conv2d):
    """Conv2d whose effective weight is W / sigma(W) (reference :49-72)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True):
        # argument order of the reference (:52-53 = nn.Conv2d's); dilated / grouped SN convolutions are nowhere on the path
        if dilation != 1 or groups != 1:
            raise NotImplementedError('SNConv2d: dilation / groups other than 1 are not on the HIP path')
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias)
        self._init_u(out_channels)

    def effective_weight(self):
        return self.W_bar

    def forward(self, input):
        return ops.conv2d(input, self.W_bar, self.bias, self.stride, self.padding)


class SNLinear(_SNMixin, nn.Module):
    """nn.Linear whose effective weight is W / sigma(W) (reference :28-47): y = x W_bar^T + b for x (..., in_features).
    The product runs on the conv kernels as a 1x1 convolution over a 1x1 plane."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        bound = 1.0 / math.sqrt(in_features)          # nn.Linear's construction-time init
        self.weight = nn.Parameter((torch.rand(out_features, in_features) * 2 - 1) * bound)
        self.bias = nn.Parameter((torch.rand(out_features) * 2 - 1) * bound) if bias else None
        self._init_u(out_features)

    def forward(self, input):
        lead = input.shape[:-1]
        x4 = input.reshape(-1, self.in_features, 1, 1)
        wb = self.W_bar
        w4 = wb.view(self.out_features, self.in_features, 1, 1)
        w4._him_wkey = id(self.weight)
        y = ops.conv2d(x4, w4, self.bias, 1, 0)
        return y.view(*lead, self.out_features)
