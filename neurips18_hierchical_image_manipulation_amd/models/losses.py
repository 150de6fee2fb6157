"""GANLoss / VGGLoss of the reference's ``models/losses.py`` on fused HIP reductions."""
import os
import torch.nn as nn

from .. import ops
from .layer_util import Vgg19


class GANLoss(nn.Module):
    """LSGAN: sum over scales of mean((logits - target)^2) on the last tensor of each scale (:40-50); ``use_lsgan=False``
    (``--no_lsgan``): nn.BCELoss on the Sigmoid outputs (:17-20; ops.bce_mean = him_bce_mean_*).  The constant target tensor
    of the reference is a fill (BCE) or never materialised (LSGAN)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=None):
        super().__init__()
        self.use_lsgan = bool(use_lsgan)
        self.real_label, self.fake_label = target_real_label, target_fake_label

    def _one(self, pred, t):
        if self.use_lsgan:
            return ops.mse_const(pred, t)
        import torch
        return ops.bce_mean(pred, torch.full_like(pred, t))

    def __call__(self, input, target_is_real):
        t = self.real_label if target_is_real else self.fake_label
        if isinstance(input[0], list):
            # loss = 0; loss += criterion(pred_i, target) per scale (:44-49): one launch instead of a chain of scalar adds
            return ops.lincomb([self._one(input_i[-1], t) for input_i in input])
        return self._one(input[-1], t)


# ReLU backward of the VGG chain folded into the kernels that produce the gradients (see Vgg19.forward); 0 = separate passes
from ..config import SCHED


class VGGLoss(nn.Module):
    def __init__(self, gpu_ids=None, normalize=False):
        super().__init__()
        if normalize:
            raise NotImplementedError('VGGLoss(normalize=True) is never used by the reference models')
        self.vgg = Vgg19()
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def target_features(self, y):
        """VGG features of the (detached) target; may be computed ahead of time on another stream."""
        import torch
        with torch.no_grad():
            return self.vgg(y)

    def forward(self, x, y, y_vgg=None):
        gated = SCHED.vgg_gated and x.requires_grad
        x_vgg = self.vgg(x, gated=gated)
        if y_vgg is None:
            y_vgg = self.target_features(y)
        return ops.l1_weighted_sum(list(zip(x_vgg, y_vgg)), self.weights, gate_relu=gated)
