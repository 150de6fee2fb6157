"""Multi-scale PatchGAN on the HIP layer executor (reference ``models/Discriminator_NET.py:11-118``; state-dict keys
``scale<i>_layer<j>.0.{weight,bias}`` with getIntermFeat, ``layer<i>.<n>.{weight,bias}`` without -- the reference builds
its discriminator with ``getIntermFeat = not opt.no_ganFeat_loss``, ``pix2pixHD_condImg_model.py:74-75``)."""
import os
import re

import torch
import torch.nn as nn

from .. import ops
from ..nn import InstanceNorm2d, BatchNorm2d, LeakyReLU, Sigmoid, FusedSequential, AvgPool3s2, _pw
from .layer_util import weights_init


def _flat_index(n_layers):
    """Position of block j's first module inside the reference's flattened ``NLayerDiscriminator.model`` Sequential
    (Discriminator_NET.py:100-104): [conv, lrelu] + (n_layers - 1) x [conv, norm, lrelu] + [conv, norm, lrelu] + [conv]."""
    lens = [2] + [3] * n_layers + [1]
    return [sum(lens[:j]) for j in range(n_layers + 2)]


def _keys_to_flat(module, state_dict, prefix, local_metadata):
    """state_dict hook (getIntermFeat=False): ``scale<i>_layer<j>.<k>.*`` -> ``layer<i>.<first[j] + k>.*``, order kept."""
    first = _flat_index(module.n_layers)
    pat = re.compile(re.escape(prefix) + r'scale(\d+)_layer(\d+)\.(\d+)\.(.*)$')
    items = list(state_dict.items())
    meta = getattr(state_dict, '_metadata', None)
    state_dict.clear()
    for k, v in items:
        m = pat.match(k)
        if m:
            k = '%slayer%s.%d.%s' % (prefix, m.group(1), first[int(m.group(2))] + int(m.group(3)), m.group(4))
        state_dict[k] = v
    if meta is not None:
        state_dict._metadata = meta
    return state_dict


def _keys_from_flat(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
    """load_state_dict pre-hook (getIntermFeat=False): the inverse renaming, in place."""
    first = _flat_index(module.n_layers)
    pat = re.compile(re.escape(prefix) + r'layer(\d+)\.(\d+)\.(.*)$')
    for k in list(state_dict.keys()):
        m = pat.match(k)
        if m:
            n = int(m.group(2))
            j = max(jj for jj, f in enumerate(first) if f <= n)
            state_dict['%sscale%s_layer%d.%d.%s' % (prefix, m.group(1), j, n - first[j], m.group(3))] = state_dict.pop(k)


def lr_control(loss_G, loss_D_real, loss_D_fake, gan_margin=0.3):
    """The reference's host-side gate (:190-211): -> (g_lr, d_lr) in {0.0, 1.0}.  D rests while either of its losses is below
    the margin, G rests while either is above 1 - margin, never both.  Reads the three scalars back (one sync); the box2mask
    trainer evaluates the same predicate on the device instead (``ops.lr_control`` / ``him_lr_control``)."""
    real, fake = float(loss_D_real.detach().reshape(-1)[0]), float(loss_D_fake.detach().reshape(-1)[0])
    update_d = not (real < gan_margin or fake < gan_margin)
    update_g = not (real > 1 - gan_margin or fake > 1 - gan_margin)
    if not (update_d or update_g):
        update_d = update_g = True
    what = 'Update Both' if update_g and update_d else ('Froze Discriminator' if update_g else 'Froze Generator')
    print('%s\t[G=%.3f],[DR=%.3f],[DF=%.3f]' % (what, float(loss_G.detach().reshape(-1)[0]), real, fake))
    return float(update_g), float(update_d)


class NLayerDiscriminator(nn.Module):
    """One PatchGAN (reference Discriminator_NET.py:60-125) as box2mask's --which_gan patch builds it: getIntermFeat False
    -> a flat ``model`` Sequential (keys ``model.<i>.*``), use_sigmoid True -> a trailing Sigmoid for nn.BCELoss;
    ``forward(input, cond)`` concatenates the condition behind the input.  (MultiscaleDiscriminator below builds its
    scales from the same blocks under its own names.)"""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='instance', use_sigmoid=False, getIntermFeat=False):
        super().__init__()
        from ..nn import Conv2d
        if getIntermFeat:
            raise NotImplementedError('NLayerDiscriminator(getIntermFeat=True) is only reached through '
                                      'MultiscaleDiscriminator in the reference; use that class')
        if norm_layer not in ('instance', 'batch'):
            raise NotImplementedError('normalization layer [%s] is not found' % norm_layer)
        norm = InstanceNorm2d if norm_layer == 'instance' else BatchNorm2d
        self.getIntermFeat, self.n_layers = False, n_layers
        seq = [Conv2d(input_nc, ndf, 4, 2, 2), LeakyReLU(0.2)]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq += [Conv2d(nf_prev, nf, 4, 2, 2), norm(nf), LeakyReLU(0.2)]
        nf_prev, nf = nf, min(nf * 2, 512)
        seq += [Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), LeakyReLU(0.2), Conv2d(nf, 1, 4, 1, 2)]
        if use_sigmoid:
            seq.append(Sigmoid())
        self.model = FusedSequential(*seq)
        self.apply(weights_init)        # reference :109

    def forward(self, input, cond=None):
        if cond is not None:
            input = ops.cat_channels([input, cond])
        return self.model(input)


class NLayerResDiscriminator(nn.Module):
    """box2mask's --which_gan patch_res (reference Discriminator_NET.py:118-183): the stride-2 stages of the PatchGAN are
    ConvResnetBlocks (kernel 4, LeakyReLU 0.2 on the stage input, conv1x1 + norm shortcut); the stride-1 block, the
    1-channel head and the Sigmoid follow in the same flat ``model`` Sequential (keys ``model.<i>.deep.1.weight`` ...)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='instance', use_sigmoid=False, getIntermFeat=False,
                 num_resnetblocks=1):
        super().__init__()
        from ..nn import Conv2d
        from .MaskTwoStreamConvSwitch_NET import ConvResnetBlock
        if getIntermFeat or num_resnetblocks != 1:
            raise NotImplementedError('NLayerResDiscriminator: getIntermFeat / num_resnetblocks > 1 have no call site upstream')
        if norm_layer not in ('instance', 'batch'):
            raise NotImplementedError('normalization layer [%s] is not found' % norm_layer)
        norm = InstanceNorm2d if norm_layer == 'instance' else BatchNorm2d
        self.getIntermFeat, self.n_layers = False, n_layers
        seq = [ConvResnetBlock(input_nc, ndf, 2, 4, norm, LeakyReLU(0.2))]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq.append(ConvResnetBlock(nf_prev, nf, 2, 4, norm, LeakyReLU(0.2)))
        nf_prev, nf = nf, min(nf * 2, 512)
        seq += [Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), LeakyReLU(0.2), Conv2d(nf, 1, 4, 1, 2)]
        if use_sigmoid:
            seq.append(Sigmoid())
        self.model = nn.Sequential(*seq)
        self.apply(weights_init)        # reference :170

    def forward(self, input, cond=None):
        h = ops.cat_channels([input, cond]) if cond is not None else input
        layers = list(self.model)
        for blk in layers[:self.n_layers]:
            h, _ = blk(h)
        from ..nn import run_layers
        return run_layers(layers[self.n_layers:], h)


class MultiscaleDiscriminator(nn.Module):
    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='instance', use_sigmoid=False, num_D=3,
                 getIntermFeat=True, spectral_norm=False):
        """``spectral_norm`` (the build's ``--sn_D``): every convolution is an ``SNConv2d`` (models/sn_utils.py; the
        reference wires its SN layers only into the unreachable Res_Discriminator) -- same keys plus ``...0.u``."""
        super().__init__()
        if spectral_norm:
            from .sn_utils import SNConv2d as Conv2d
        else:
            from ..nn import Conv2d
        if norm_layer not in ('instance', 'batch'):
            raise NotImplementedError('normalization layer [%s] is not found' % norm_layer)
        norm = InstanceNorm2d if norm_layer == 'instance' else BatchNorm2d   # 'batch': the box2mask discriminator
        if use_sigmoid and getIntermFeat:
            # The reference cannot run this either: MultiscaleDiscriminator copies only model0..model<n_layers+1> of each
            # NLayerDiscriminator when getIntermFeat is set (Discriminator_NET.py:24-27), which drops the trailing Sigmoid,
            # so GANLoss's nn.BCELoss (losses.py:19-20) is fed raw logits (an error in torch >= 0.4, NaNs before).
            # With --no_ganFeat_loss the Sigmoid sits inside ``layer<i>`` (:27-28, :95-96) and the flag works (round 6).
            raise NotImplementedError('--no_lsgan needs --no_ganFeat_loss: with feature matching the reference drops the '
                                      'Sigmoid in front of its BCELoss (Discriminator_NET.py:24-27)')
        self.num_D, self.n_layers, self.getIntermFeat = num_D, n_layers, bool(getIntermFeat)
        if not getIntermFeat:
            # --no_ganFeat_loss: the reference keeps each scale as ONE flattened Sequential ``layer<i>`` (:27-28) -- the same
            # convolutions on the same inputs, other checkpoint keys.  The blocks stay separate here (one executor for both
            # forms); only the names a checkpoint sees change.
            self._register_state_dict_hook(_keys_to_flat)
            self._register_load_state_dict_pre_hook(_keys_from_flat, with_module=True)
        for i in range(num_D):
            blocks = [[Conv2d(input_nc, ndf, 4, 2, 2), LeakyReLU(0.2)]]
            nf = ndf
            for _ in range(1, n_layers):
                nf_prev, nf = nf, min(nf * 2, 512)
                blocks.append([Conv2d(nf_prev, nf, 4, 2, 2), norm(nf), LeakyReLU(0.2)])
            nf_prev, nf = nf, min(nf * 2, 512)
            blocks.append([Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), LeakyReLU(0.2)])
            blocks.append([Conv2d(nf, 1, 4, 1, 2)] + ([Sigmoid()] if use_sigmoid else []))     # reference :95-96
            for j, b in enumerate(blocks):
                setattr(self, 'scale%d_layer%d' % (i, j), FusedSequential(*b))
        self.downsample = AvgPool3s2()
        self.apply(weights_init)        # reference :34 (conv N(0, .02); BatchNorm weight N(1, .02), bias 0)

    def _scale(self, i, x):
        feats, h = [], x
        for j in range(self.n_layers + 2):
            layer = getattr(self, 'scale%d_layer%d' % (self.num_D - 1 - i, j))   # :51 index reversal
            if j == 0 and isinstance(h, tuple):
                # (condition, image) kept apart (ops.CondImage): layer 0 is Conv2d + LeakyReLU
                conv, act = layer[0], layer[1]
                h = ops.cond_image_conv2d(h[0], h[1], conv.effective_weight(), _pw(conv.bias), conv.stride, conv.padding,
                                          act.act, act.slope)
            else:
                h = layer(h)
            feats.append(h)
        return feats

    def _forward_split(self, ci):
        """``ops.CondImage`` input: the pooled condition comes from the per-step cache, only the image is pooled here."""
        conds = ops.cond_pyramid(ci.cond, self.num_D)
        result, img = [], ci.image
        for i in range(self.num_D):
            result.append(self._scale(i, (conds[i], img)))
            if i != self.num_D - 1:
                img = self.downsample(img)
        return result

    def forward(self, input):
        if isinstance(input, ops.CondImage):
            if SCHED.d_scale_streams and self.num_D > 1:
                input = input.cat()
            else:
                return self._forward_split(input)
        if SCHED.d_scale_streams and input.is_cuda and self.num_D > 1:
            return self._forward_streams(input)
        result, x = [], input
        for i in range(self.num_D):
            result.append(self._scale(i, x))
            if i != self.num_D - 1:
                x = self.downsample(x)
        return result

    def _forward_streams(self, input):
        """The scales are independent given their (pooled) inputs: scales 1.. run on streams of their own next to scale 0
        (their launches have few tiles -- 1/4, 1/16 of the pixels -- and leave most of the chip idle when run alone).
        Autograd replays every node on the stream its forward ran on, so the backward overlaps the same way."""
        dev = input.device
        main = torch.cuda.current_stream(dev)
        xs = [input]
        for i in range(1, self.num_D):
            xs.append(self.downsample(xs[-1]))
        result = [None] * self.num_D
        streams = []
        for i in range(self.num_D - 1, 0, -1):      # smallest first: they start while scale 0 is being enqueued
            st = _scale_stream(dev, i)
            st.wait_stream(main)
            xs[i].record_stream(st)
            with torch.cuda.stream(st):
                result[i] = self._scale(i, xs[i])
            for t in result[i]:
                t.record_stream(main)
            streams.append(st)
        result[0] = self._scale(0, xs[0])
        for st in streams:
            main.wait_stream(st)
        return result


from ..config import SCHED     # noqa: E402
_STREAMS = {}


def _scale_stream(device, i):
    s = _STREAMS.get((device, i))
    if s is None:
        s = _STREAMS[(device, i)] = torch.cuda.Stream(device=device)
    return s
