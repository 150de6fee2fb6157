"""Shared pieces of the reference's ``models/layer_util.py`` that sit on the mask2image path:
``weights_init`` (:9-16), ``get_norm_layer`` (:19-26), ``ResnetBlock`` (:333-378), ``Vgg19`` (:380-411)."""
import torch
import torch.nn as nn

from ..nn import Conv2d, InstanceNorm2d, BatchNorm2d, ReLU, MaxPool, ResnetBlock, FusedSequential  # noqa: F401


def weights_init(m, conv_sigma=0.02, bnorm_sigma=0.02):
    """reference :9-16: conv weights ~ N(0, conv_sigma); BatchNorm2d weight ~ N(1, bnorm_sigma), bias 0.  Applied by the
    reference to the pix2pixHD generator / discriminator and to every discriminator class (``self.apply(weights_init)``)."""
    name = m.__class__.__name__
    with torch.no_grad():
        if name.find('Conv') != -1 and hasattr(m, 'weight'):
            m.weight.normal_(0.0, conv_sigma)
        elif name.find('BatchNorm2d') != -1:
            m.weight.normal_(1.0, bnorm_sigma)
            m.bias.fill_(0)


def torch_default_init(net):
    """The reference never calls ``weights_init`` on the box2mask generator (models/TwoStreamAE_mask.py builds
    MaskTwoStreamConvSwitch_NET and uses it as constructed), so its layers keep ``torch.nn``'s construction-time
    initialisation: conv / transposed-conv weights ~ U(+-1/sqrt(fan_in)) (kaiming_uniform with a = sqrt(5)), fan_in =
    weight.size(1) * k * k, biases from the same bound; BatchNorm weight 1, bias 0."""
    import math
    with torch.no_grad():
        for m in net.modules():
            if m.__class__.__name__ in ('Conv2d', 'ConvTranspose2d', '_BiasFreeConv3x3') and hasattr(m, 'weight'):
                bound = 1.0 / math.sqrt(m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])
                m.weight.uniform_(-bound, bound)
                if getattr(m, 'bias', None) is not None:
                    m.bias.uniform_(-bound, bound)
    return net


def print_network(net):
    """Module + its parameter count on stdout (reference :28-35)."""
    if isinstance(net, list):
        net = net[0]
    print(net)
    print('Total number of parameters: %d' % sum(p.numel() for p in net.parameters()))


def get_norm_layer(norm_type='instance'):
    """reference :19-26: 'instance' -> InstanceNorm2d(affine=False), 'batch' -> BatchNorm2d(affine=True); anything else
    raises, as there."""
    if norm_type == 'instance':
        return InstanceNorm2d
    if norm_type == 'batch':
        return BatchNorm2d
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


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
                # the weights never change: descriptors of these layers carry HIM_ALGO_FROZEN_WEIGHTS (Winograd
                # F(4x4,3x3) for the >= 256-channel convolutions, panels built once per run)
                p._him_frozen = True

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

    def forward(self, X, gated=False):
        """[relu1_1, relu2_1, relu3_1, relu4_1, relu5_1].  ``gated`` (used by VGGLoss only, which owns every consumer of
        the five outputs): each ReLU's backward is applied by whoever produces the gradient of its output -- the next
        conv's data-gradient epilogue, the pool's backward, the L1 backward -- so no stand-alone activation-backward pass
        runs over the 13 feature maps.  The caller MUST gate the gradients it sends into the outputs
        (``ops.l1_weighted_sum(..., gate_relu=True)``)."""
        from ..nn import run_layers
        out, h = [], X
        state = {'prev_relu': False} if gated else None
        for k in range(5):
            h = run_layers(list(getattr(self, 'slice%d' % (k + 1))), h, relu_gated=state)
            out.append(h)
        return out
