"""Shared pieces of the reference's ``models/layer_util.py`` that sit on the mask2image path:
``weights_init`` (:9-16), ``get_norm_layer`` (:19-26), ``ResnetBlock`` (:333-378), ``Vgg19`` (:380-411)."""
import torch
import torch.nn as nn

from ..nn import Conv2d, InstanceNorm2d, ReLU, MaxPool, ResnetBlock, FusedSequential  # noqa: F401


def weights_init(m, conv_sigma=0.02):
    if m.__class__.__name__.find('Conv') != -1 and hasattr(m, 'weight'):
        with torch.no_grad():
            m.weight.normal_(0.0, conv_sigma)


def get_norm_layer(norm_type='instance'):
    if norm_type == 'instance':
        return InstanceNorm2d
    raise NotImplementedError('normalization layer [%s] is not found on the HIP path' % norm_type)


VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]
_SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]


class Vgg19(nn.Module):
    """torchvision ``vgg19().features[0:30]`` cut into relu1_1..relu5_1; keys ``slice<k>.<idx>.*`` (a
    torchvision ``vgg19-*.pth`` maps onto them with ``load_torchvision_state_dict``).  Weights are frozen.
    There is no network here: without a user-supplied file the weights are the build's seeded synthetic ones."""

    def __init__(self, requires_grad=False):
        super().__init__()
        layers, c = [], 3
        for v in VGG19_CFG:
            if v == 'M':
                layers.append(MaxPool(2))
            else:
                layers += [Conv2d(c, v, 3, padding=1), ReLU()]
                c = v
        for k, (a, b) in enumerate(_SLICES):
            seq = FusedSequential()
            for idx in range(a, b):
                seq.add_module(str(idx), layers[idx])
            setattr(self, 'slice%d' % (k + 1), seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_state_dict(self, sd):
        """Accepts torchvision's ``features.<idx>.{weight,bias}`` naming."""
        own = self.state_dict()
        mapped = {}
        for k, (a, b) in enumerate(_SLICES):
            for idx in range(a, b):
                for s in ('weight', 'bias'):
                    src = 'features.%d.%s' % (idx, s)
                    dst = 'slice%d.%d.%s' % (k + 1, idx, s)
                    if src in sd and dst in own:
                        mapped[dst] = sd[src]
        self.load_state_dict(mapped)

    def forward(self, X):
        out, h = [], X
        for k in range(5):
            h = getattr(self, 'slice%d' % (k + 1))(h)
            out.append(h)
        return out
