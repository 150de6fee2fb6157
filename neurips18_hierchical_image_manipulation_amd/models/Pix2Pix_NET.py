"""Generators of the mask2image path on the HIP layer executor.

Mirrors reference ``models/Pix2Pix_NET.py``: GlobalGenerator (:63-101), LocalEnhancer (:8-61),
GlobalTwoStreamGenerator (:103-247, 'early_add' fusion).  Layer lists are index-compatible with the
reference ``nn.Sequential``s so published ``*_net_G.pth`` files load unchanged.
"""
import torch.nn as nn

from .. import ops
from ..nn import (Conv2d, ConvTranspose2d, ReflectionPad2d, InstanceNorm2d, ReLU, Tanh, ResnetBlock,
                  FusedSequential, AvgPool3s2, run_layers)


def stem(cin, ngf):
    return [ReflectionPad2d(3), Conv2d(cin, ngf, 7), InstanceNorm2d(ngf), ReLU()]


def down(c):
    return [Conv2d(c, 2 * c, 3, stride=2, padding=1), InstanceNorm2d(2 * c), ReLU()]


def up(cin, cout):
    return [ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1), InstanceNorm2d(cout), ReLU()]


def head(ngf, out_nc):
    return [ReflectionPad2d(3), Conv2d(ngf, out_nc, 7), Tanh()]


def _check_norm(norm_layer):
    if norm_layer != 'instance':
        raise NotImplementedError('normalization layer [%s] is not on the HIP path (instance only)' % norm_layer)


class GlobalGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer='instance',
                 padding_type='reflect', use_output_gate=False):
        assert n_blocks >= 0
        super().__init__()
        _check_norm(norm_layer)
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        self.input_nc, self.output_nc, self.use_output_gate = input_nc, output_nc, use_output_gate
        seq = stem(input_nc, ngf)
        for i in range(n_downsampling):
            seq += down(ngf * 2 ** i)
        seq += [ResnetBlock(ngf * 2 ** n_downsampling) for _ in range(n_blocks)]
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            seq += up(c, c // 2)
        seq += head(ngf, output_nc)
        self.model = FusedSequential(*seq)

    def forward(self, input, mask=None):
        out = self.model(input)
        if self.use_output_gate and mask is not None:
            # (1-mask)*input[:, -3:] + mask*out, reading the image straight out of the G input buffer
            out = ops.blend(input, out, mask, a0=self.input_nc - 3)
        return out


class LocalEnhancer(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9,
                 n_local_enhancers=1, n_blocks_local=3, norm_layer='instance', padding_type='reflect'):
        super().__init__()
        _check_norm(norm_layer)
        self.n_local_enhancers = n_local_enhancers
        g = GlobalGenerator(input_nc, output_nc, ngf * 2 ** n_local_enhancers, n_downsample_global,
                            n_blocks_global, norm_layer).model
        self.model = FusedSequential(*list(g.children())[:-3])
        for n in range(1, n_local_enhancers + 1):
            c = ngf * 2 ** (n_local_enhancers - n)
            dn = stem(input_nc, c) + down(c)
            upl = [ResnetBlock(2 * c) for _ in range(n_blocks_local)] + up(2 * c, c)
            if n == n_local_enhancers:
                upl += head(ngf, output_nc)
            setattr(self, 'model%d_1' % n, FusedSequential(*dn))
            setattr(self, 'model%d_2' % n, FusedSequential(*upl))
        self.downsample = AvgPool3s2()

    def forward(self, input):
        pyr = [input]
        for _ in range(self.n_local_enhancers):
            pyr.append(self.downsample(pyr[-1]))
        out = self.model(pyr[-1])
        for n in range(1, self.n_local_enhancers + 1):
            xi = pyr[self.n_local_enhancers - n]
            out = getattr(self, 'model%d_2' % n)(ops.add(getattr(self, 'model%d_1' % n)(xi), out))
        return out


class GlobalTwoStreamGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer='instance',
                 padding_type='reflect', use_skip=False, which_stream='ctx', use_output_gate=False,
                 feat_fusion='early_add', extra_embed=False):
        assert n_blocks >= 0
        super().__init__()
        _check_norm(norm_layer)
        if feat_fusion != 'early_add':
            raise NotImplementedError('feat_fusion [%s]: only early_add (the shipped recipe) is on the HIP path'
                                      % feat_fusion)
        if use_skip and 'ctx' not in which_stream:
            # the skips are the CONTEXT encoder's features (reference :232-241); with --which_encoder label the reference
            # builds a decoder with doubled inputs and fails inside it at the first forward (:225)
            raise NotImplementedError('--use_skip needs the context stream (--which_encoder ctx | ctx_label)')
        self.nd, self.use_skip, self.which_stream = n_downsampling, use_skip, which_stream
        self.use_output_gate, self.output_nc = use_output_gate, output_nc
        self.feat_dim = ngf * 2 ** n_downsampling

        def downs():
            seq = []
            for i in range(n_downsampling):
                seq += down(ngf * 2 ** i)
            return FusedSequential(*seq)

        if 'ctx' in which_stream:
            self.ctx_inputEmbedder = FusedSequential(*stem(6 if extra_embed else 3, ngf))
            self.ctx_downsampler = downs()
        if 'label' in which_stream:
            self.obj_inputEmbedder = FusedSequential(*stem(input_nc, ngf))
            self.obj_downsampler = downs()
        self.latent_embedder = FusedSequential(*[ResnetBlock(self.feat_dim) for _ in range(n_blocks)])
        dec = []
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            dec += up(2 * c if (use_skip and i > 0) else c, c // 2)
        self.decoder = FusedSequential(*dec)
        self.outputEmbedder = FusedSequential(*head(ngf, output_nc))

    def _encode(self, embedder, downsampler, x, want_skips):
        h, skips = embedder(x), []
        layers = list(downsampler)
        for s in range(self.nd):                       # one fused conv3s2-IN-ReLU per stage
            h = run_layers(layers[3 * s:3 * s + 3], h)
            if want_skips and s < self.nd - 1:          # reference: i%3==2 and i < 3*nd-1
                skips.append(h)
        return h, skips

    def forward(self, img, label, mask):
        ctx = obj = None
        skips = []
        if 'ctx' in self.which_stream:
            ctx, skips = self._encode(self.ctx_inputEmbedder, self.ctx_downsampler, img, self.use_skip)
        if 'label' in self.which_stream:
            obj, _ = self._encode(self.obj_inputEmbedder, self.obj_downsampler, label, False)
        if self.which_stream == 'ctx_label':
            m = ops.maxpool(mask, 2 ** self.nd)
            h = ops.blend(ctx, obj, m)                 # (1-m)*ctx + m*obj
        elif self.which_stream == 'ctx':
            h = ctx
        else:
            h = obj
        h = self.latent_embedder(h)
        layers = list(self.decoder)
        for s in range(self.nd):
            if self.use_skip and skips and s > 0:
                h = ops.cat_channels([skips[-s], h])   # encoder channels first
            h = run_layers(layers[3 * s:3 * s + 3], h)
        out = self.outputEmbedder(h)
        if self.use_output_gate:
            out = ops.blend(img, out, mask, a0=0)      # img[:, :3]
        return out
