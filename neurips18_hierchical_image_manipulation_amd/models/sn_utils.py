"""Spectral normalisation of the reference's ``models/sn_utils.py`` on the HIP power-iteration op.
(Unused by any live reference model -- shipped as a standalone op + layer, parity pinned at function level.)"""
import torch
import torch.nn as nn

from .. import ops
from ..nn import Conv2d


def max_singular_value(W, u=None, Ip=1):
    if Ip != 1:
        raise NotImplementedError('Ip=1 only (the reference default)')
    return ops.sn_max_singular_value(W, u)


class SNConv2d(Conv2d):
    """Conv2d whose effective weight is W / sigma(W); ``u`` is persisted while training (:62-67)."""
    Ip = 1

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, k, stride, padding, bias)
        self.u = nn.Parameter(torch.randn(1, cout), requires_grad=False)

    @property
    def W_bar(self):
        sigma, _u = max_singular_value(self.weight, self.u, self.Ip)
        if self.training:
            with torch.no_grad():
                self.u.copy_(_u)
        return ops.div_scalar(self.weight, sigma)

    def effective_weight(self):
        return self.W_bar

    def forward(self, x):
        return ops.conv2d(x, self.W_bar, self.bias, self.stride, self.padding)
