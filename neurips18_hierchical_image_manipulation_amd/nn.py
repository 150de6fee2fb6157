"""Parameter-holding layer modules and the fusing executor.

The layer classes are *markers* that carry parameters under exactly the reference's state_dict keys
(``model.1.weight`` ...).  They are never run one by one: ``run_layers`` walks a layer list and fuses
    [ReflectionPad2d] Conv2d|ConvTranspose2d [InstanceNorm2d] [ReLU|LeakyReLU|Tanh]
into one conv kernel (reflect indexing in the tile loader, bias + activation in the epilogue when no norm
sits in between) plus, when a norm is present, one fused InstanceNorm+activation(+residual) kernel.
"""
import contextlib
import math
import os

import torch
import torch.nn as nn

from . import ops


class _Frozen(object):
    on = False


from .config import SCHED     # SCHED.dead_bias_skip = False computes the dead bias gradients anyway


@contextlib.contextmanager
def frozen_params():
    """Run modules with detached weights: gradients flow to the input only (the D pass inside loss_G --
    the reference computes those weight gradients and throws them away at train_mask2image.py:84)."""
    prev, _Frozen.on = _Frozen.on, True
    try:
        yield
    finally:
        _Frozen.on = prev


def _pw(p):
    return p.detach() if (_Frozen.on and p is not None) else p


class Conv2d(nn.Module):
    transposed = False

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        self.k, self.stride, self.padding = k, stride, padding
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k) * 0.02)
        bound = 1.0 / math.sqrt(cin * k * k)
        self.bias = nn.Parameter((torch.rand(cout) * 2 - 1) * bound) if bias else None

    def effective_weight(self):
        return _pw(self.weight)

    def extra_repr(self):
        return '%d->%d k%d s%d p%d' % (self.weight.shape[1], self.weight.shape[0], self.k, self.stride, self.padding)


class ConvTranspose2d(nn.Module):
    transposed = True

    def __init__(self, cin, cout, k, stride=2, padding=1, output_padding=1, bias=True):
        super().__init__()
        self.k, self.stride, self.padding, self.output_padding = k, stride, padding, output_padding
        self.weight = nn.Parameter(torch.randn(cin, cout, k, k) * 0.02)
        bound = 1.0 / math.sqrt(cout * k * k)
        self.bias = nn.Parameter((torch.rand(cout) * 2 - 1) * bound) if bias else None

    def effective_weight(self):
        return _pw(self.weight)


class ReflectionPad2d(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p


class InstanceNorm2d(nn.Module):
    def __init__(self, ch, eps=1e-5):
        super().__init__()
        self.ch, self.eps = ch, eps

    def apply_to(self, x, act='none', slope=0.0, residual=None):
        """act(InstanceNorm2d(x)) [+ residual] -- same call shape as BatchNorm2d.apply_to (box2mask blocks take either)."""
        return ops.instance_norm(x, residual, act, slope, self.eps)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d(affine=True) parameter / buffer holder (state_dict keys weight, bias, running_mean, running_var,
    num_batches_tracked); applied through ``ops.batch_norm`` with the module's train/eval mode."""

    def __init__(self, ch, eps=1e-5, momentum=0.1):
        super().__init__()
        self.ch, self.eps, self.momentum = ch, eps, momentum
        self.weight = nn.Parameter(torch.ones(ch))
        self.bias = nn.Parameter(torch.zeros(ch))
        self.register_buffer('running_mean', torch.zeros(ch))
        self.register_buffer('running_var', torch.ones(ch))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        # checkpoints written by torch < 0.4.1 (the reference's era) carry no ``num_batches_tracked``: keep the counter
        key = prefix + 'num_batches_tracked'
        if key not in state_dict:
            state_dict = dict(state_dict)
            state_dict[key] = self.num_batches_tracked.detach().clone()
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def apply_to(self, x, act='none', slope=0.0, residual=None):
        if self.training:
            self.num_batches_tracked += 1
        return ops.batch_norm(x, _pw(self.weight), _pw(self.bias), self.running_mean, self.running_var, self.training,
                              self.momentum, self.eps, act, slope, residual)


class ReLU(nn.Module):
    act, slope = 'relu', 0.0


class LeakyReLU(nn.Module):
    act = 'lrelu'

    def __init__(self, slope=0.2):
        super().__init__()
        self.slope = slope


class Tanh(nn.Module):
    act, slope = 'tanh', 0.0


class Sigmoid(nn.Module):
    act, slope = 'sigmoid', 0.0


_ACTS = (ReLU, LeakyReLU, Tanh, Sigmoid)


def run_layers(layers, x, final_residual=None, relu_gated=None):
    """Execute a list of marker layers / sub-modules on x with the fusion described in the module docstring.
    ``final_residual`` is added by the LAST InstanceNorm of the list (ResnetBlock tail).
    ``relu_gated`` (a dict carrying ``prev_relu`` across calls): the list is a conv -> ReLU [-> MaxPool] chain whose
    caller gates every gradient it feeds back into the chain's outputs (VGGLoss); each ReLU backward then rides in the
    kernel that produces the gradient (next conv's data gradient / the pool's backward) instead of its own pass."""
    layers = list(layers)
    n, i = len(layers), 0
    last_norm = max([j for j, l in enumerate(layers) if isinstance(l, (InstanceNorm2d, BatchNorm2d))], default=-1)
    while i < n:
        l = layers[i]
        pad_mode, rpad = 'zero', 0
        if isinstance(l, ReflectionPad2d):
            if i + 1 >= n or not isinstance(layers[i + 1], Conv2d):
                raise ValueError('ReflectionPad2d must be followed by Conv2d')
            pad_mode, rpad = 'reflect', l.p
            i += 1
            l = layers[i]
        if isinstance(l, (Conv2d, ConvTranspose2d)):
            norm = act = None
            norm_idx = -2
            j = i + 1
            if j < n and isinstance(layers[j], (InstanceNorm2d, BatchNorm2d)):
                norm, norm_idx = layers[j], j
                j += 1
            if j < n and isinstance(layers[j], _ACTS):
                act = layers[j]
                j += 1
            aname = act.act if act is not None else 'none'
            slope = getattr(act, 'slope', 0.0) if act is not None else 0.0
            epi = 'none' if norm is not None else aname
            w, b = l.effective_weight(), _pw(l.bias)
            if b is not None and SCHED.dead_bias_skip and (isinstance(norm, InstanceNorm2d) or
                                                      (isinstance(norm, BatchNorm2d) and norm.training)):
                # A bias in front of a normalisation that subtracts the plane (batch) mean cancels exactly: its true
                # gradient is identically zero and what autograd would produce is rounding noise (1e-9 of the net's
                # gradients; the reference's Adam turns that noise into a random walk of a parameter the output does not
                # depend on).  The bias still takes part in the forward (same rounding as the reference); its gradient
                # pass -- a full read of dy per layer -- is skipped and the parameter stays where it is.
                l.bias._him_dead_grad = True
                b = b.detach()
            elif b is not None and getattr(l.bias, '_him_dead_grad', False):
                # the same conv now runs WITHOUT a mean-subtracting norm behind it (BatchNorm switched to eval()): its
                # bias gradient is live again and the data-parallel reducer must count it (dist.GradReducer.begin)
                l.bias._him_dead_grad = False
            if l.transposed:
                x = ops.conv_transpose2d(x, w, b, l.stride, l.padding, l.output_padding, epi, slope)
            else:
                if pad_mode == 'reflect' and l.padding != 0:
                    raise ValueError('reflect pad + conv padding')
                if relu_gated is not None and norm is None and epi == 'relu':
                    x = ops.conv2d(x, w, b, l.stride, rpad if pad_mode == 'reflect' else l.padding, pad_mode, epi, slope,
                                   grad_premasked=True, gate_dx=relu_gated.get('prev_relu', False))
                    relu_gated['prev_relu'] = True
                else:
                    if relu_gated is not None:
                        raise ValueError('relu_gated chains hold conv -> ReLU [-> MaxPool] only')
                    if isinstance(norm, InstanceNorm2d):
                        # Conv2d -> InstanceNorm -> activation block as one op (ops.conv2d_in_act)
                        res = final_residual if (final_residual is not None and norm_idx == last_norm) else None
                        if res is not None and act is not None:
                            raise ValueError('residual after an activated norm is not a ResnetBlock tail')
                        x = ops.conv2d_in_act(x, w, b, l.stride, rpad if pad_mode == 'reflect' else l.padding, pad_mode,
                                              norm.eps, aname, slope, res)
                        norm = None
                    else:
                        x = ops.conv2d(x, w, b, l.stride, rpad if pad_mode == 'reflect' else l.padding, pad_mode, epi, slope)
            if norm is not None:
                res = final_residual if (final_residual is not None and norm_idx == last_norm) else None
                if res is not None and act is not None:
                    raise ValueError('residual after an activated norm is not a ResnetBlock tail')
                if isinstance(norm, BatchNorm2d):
                    x = norm.apply_to(x, aname, slope, res)
                else:
                    x = ops.instance_norm(x, res, aname, slope, norm.eps)
            i = j
        elif isinstance(l, InstanceNorm2d):
            x = ops.instance_norm(x, None, 'none', 0.0, l.eps)
            i += 1
        elif isinstance(l, BatchNorm2d):
            x = l.apply_to(x)
            i += 1
        elif isinstance(l, _ACTS):
            raise ValueError('stand-alone activation is not on the hot path')
        elif relu_gated is not None:
            if not isinstance(l, MaxPool) or not relu_gated.get('prev_relu', False):
                raise ValueError('relu_gated chains hold conv -> ReLU [-> MaxPool] only')
            x = ops.maxpool(x, l.k, relu_gate=True)
            relu_gated['prev_relu'] = False     # the pool's output gradient needs no gate
            i += 1
        else:
            x = l(x)
            i += 1
    return x


class FusedSequential(nn.Sequential):
    def forward(self, x):
        return run_layers(list(self), x)


class ResnetBlock(nn.Module):
    """x + N(conv3(refpad(ReLU(N(conv3(refpad(x)))))))  (reference models/layer_util.py:333-378), N = InstanceNorm2d or,
    under --norm batch, BatchNorm2d(affine=True); convolutions at ``conv_block.1`` / ``conv_block.5``, BatchNorm
    parameters and statistics at ``conv_block.2`` / ``conv_block.6``."""

    def __init__(self, dim, padding_type='reflect', norm_layer=None, activation=None, use_dropout=False):
        """Constructor arguments of the reference (:334-335).  Every call site there passes reflection padding, the layer
        of ``get_norm_layer`` and ``nn.ReLU(True)`` and leaves ``use_dropout`` False (Pix2Pix_NET.py:32,84,173;
        MaskTwoStreamConv*_NET.py: ``use_dropout=False``; the parsers' --use_dropout reaches no constructor), so the
        dropout form is unreachable from the reference's entry points: it fails here instead of being computed as the
        plain block."""
        super().__init__()
        if padding_type != 'reflect':
            raise NotImplementedError('ResnetBlock: padding [%s] is not on the HIP path (reflect only)' % padding_type)
        if use_dropout:
            raise NotImplementedError('ResnetBlock: use_dropout is not on the HIP path (no reference call site sets it)')
        norm_layer = InstanceNorm2d if norm_layer is None else norm_layer
        if not isinstance(norm_layer(dim), (InstanceNorm2d, BatchNorm2d)):
            raise NotImplementedError('ResnetBlock: norm_layer must come from get_norm_layer("instance" | "batch")')
        if activation is not None and not isinstance(activation, (ReLU, nn.ReLU)):
            raise NotImplementedError('ResnetBlock: activation must be ReLU')
        self.conv_block = nn.Sequential(ReflectionPad2d(1), Conv2d(dim, dim, 3), norm_layer(dim), ReLU(),
                                        ReflectionPad2d(1), Conv2d(dim, dim, 3), norm_layer(dim))

    def forward(self, x):
        cb = self.conv_block
        c1, c2 = cb[1], cb[5]
        if (SCHED.dead_bias_skip and type(c1) is Conv2d and type(c2) is Conv2d and not _Frozen.on
                and type(cb[2]) is InstanceNorm2d and type(cb[6]) is InstanceNorm2d and cb[2].eps == cb[6].eps
                and type(cb[3]) is ReLU and ops.resblock_supported(x, c1.weight, c2.weight)):
            # the 1024-channel stack: both InstanceNorms ride in the Winograd transforms of the two convolutions
            for c in (c1, c2):
                if c.bias is not None:
                    c.bias._him_dead_grad = True
            return ops.resnet_block(x, c1.weight, c1.bias, c2.weight, c2.bias, cb[2].eps)
        return run_layers(list(cb), x, final_residual=x)


class AvgPool3s2(nn.Module):
    def forward(self, x):
        return ops.avgpool3s2(x)


class MaxPool(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x):
        return ops.maxpool(x, self.k)
