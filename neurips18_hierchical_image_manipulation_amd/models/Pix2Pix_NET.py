"""Generators of the mask2image path on the HIP layer executor.

Mirrors reference ``models/Pix2Pix_NET.py``: GlobalGenerator (:63-101), LocalEnhancer (:8-61),
GlobalTwoStreamGenerator (:103-247, --feat_fusion early_add | early_concat | late_add | late_concat), each with
--norm instance | batch (``get_norm_layer``, models/layer_util.py:19-26).  Layer lists are index-compatible with the
reference ``nn.Sequential``s so published ``*_net_G.pth`` files load unchanged.
"""
import torch.nn as nn

from .. import ops
from ..nn import (Conv2d, ConvTranspose2d, ReflectionPad2d, InstanceNorm2d, ReLU, Tanh, ResnetBlock,
                  FusedSequential, AvgPool3s2, run_layers)
from .layer_util import get_norm_layer


def stem(cin, ngf, norm=InstanceNorm2d):
    return [ReflectionPad2d(3), Conv2d(cin, ngf, 7), norm(ngf), ReLU()]


def down(c, norm=InstanceNorm2d):
    return [Conv2d(c, 2 * c, 3, stride=2, padding=1), norm(2 * c), ReLU()]


def up(cin, cout, norm=InstanceNorm2d):
    return [ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1), norm(cout), ReLU()]


def head(ngf, out_nc):
    return [ReflectionPad2d(3), Conv2d(ngf, out_nc, 7), Tanh()]


class FeatureFusionBlock(nn.Module):
    """reference models/layer_util.py:295-330 at its one call site (Pix2Pix_NET.py:135-136: ``main_module`` an Identity).
    'add': x + y.  'concat': norm(conv1x1(ReLU(cat(x, y)))), parameters at ``conv1`` (+ ``norm1`` under --norm batch).
    ``forward(x, y)`` is the reference's call.  The generator passes the two stream features UNMASKED plus the pooled mask
    (``forward(ctx, obj, m)``) and gets the fusion of ((1-m)*ctx, m*obj) (reference Pix2Pix_NET.py:215-217): the masks
    ride in the kernels that would copy anyway."""

    def __init__(self, planes, fusion_type, norm=InstanceNorm2d):
        super().__init__()
        if fusion_type not in ('add', 'concat'):            # the reference's assert (:300)
            raise AssertionError('fusion_type [%s] must be add or concat' % fusion_type)
        self.fusion_type = fusion_type
        if fusion_type == 'concat':
            self.conv1 = Conv2d(2 * planes, planes, 1)
            self.norm1 = norm(planes)

    def forward(self, ctx, obj, m=None):
        if self.fusion_type == 'add':
            return ops.add(ctx, obj) if m is None else ops.blend(ctx, obj, m)   # (1-m)*ctx + m*obj in one pass
        h = ops.cat_channels([ctx, obj]) if m is None else ops.cat_channels([ctx, obj], m, (2, 1))
        return run_layers([self.conv1, self.norm1], ops.activation(h, 'relu'))   # cat((1-m)*ctx, m*obj) -> ReLU -> ...


class GlobalGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer='instance',
                 padding_type='reflect', use_output_gate=False):
        assert n_blocks >= 0
        super().__init__()
        norm = get_norm_layer(norm_layer)
        if padding_type != 'reflect':
            raise NotImplementedError('padding [%s] is not implemented' % padding_type)
        self.input_nc, self.output_nc, self.use_output_gate = input_nc, output_nc, use_output_gate
        seq = stem(input_nc, ngf, norm)
        for i in range(n_downsampling):
            seq += down(ngf * 2 ** i, norm)
        seq += [ResnetBlock(ngf * 2 ** n_downsampling, norm_layer=norm) for _ in range(n_blocks)]
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            seq += up(c, c // 2, norm)
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
        norm = get_norm_layer(norm_layer)
        self.n_local_enhancers = n_local_enhancers
        g = GlobalGenerator(input_nc, output_nc, ngf * 2 ** n_local_enhancers, n_downsample_global,
                            n_blocks_global, norm_layer).model
        self.model = FusedSequential(*list(g.children())[:-3])
        for n in range(1, n_local_enhancers + 1):
            c = ngf * 2 ** (n_local_enhancers - n)
            dn = stem(input_nc, c, norm) + down(c, norm)
            upl = [ResnetBlock(2 * c, norm_layer=norm) for _ in range(n_blocks_local)] + up(2 * c, c, norm)
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
        # the reference's asserts (:108-109): late fusion needs both streams
        assert not ('label' not in which_stream and 'late' in feat_fusion)
        assert not ('ctx' not in which_stream and 'late' in feat_fusion)
        super().__init__()
        norm = get_norm_layer(norm_layer)
        if feat_fusion not in ('early_add', 'early_concat', 'late_add', 'late_concat'):
            # anything else builds no latent embedder / trips FeatureFusionBlock's assert in the reference
            raise NotImplementedError('feat_fusion [%s] is not one of early_add | early_concat | late_add | late_concat'
                                      % feat_fusion)
        if use_skip and 'ctx' not in which_stream:
            # the skips are the CONTEXT encoder's features (reference :232-241); with --which_encoder label the reference
            # builds a decoder with doubled inputs and fails inside it at the first forward (:225)
            raise NotImplementedError('--use_skip needs the context stream (--which_encoder ctx | ctx_label)')
        self.nd, self.use_skip, self.which_stream = n_downsampling, use_skip, which_stream
        self.use_output_gate, self.output_nc, self.feat_fusion = use_output_gate, output_nc, feat_fusion
        self.feat_dim = ngf * 2 ** n_downsampling

        def downs():
            seq = []
            for i in range(n_downsampling):
                seq += down(ngf * 2 ** i, norm)
            return FusedSequential(*seq)

        def embedder(n):                                # reference get_embedder (:166-174)
            return FusedSequential(*[ResnetBlock(self.feat_dim, norm_layer=norm) for _ in range(n)])

        # module order = the reference's (:126-146): it is the order of the checkpoint's keys
        if 'ctx' in which_stream:
            self.ctx_inputEmbedder = FusedSequential(*stem(6 if extra_embed else 3, ngf, norm))
            self.ctx_downsampler = downs()
        if 'label' in which_stream:
            self.obj_inputEmbedder = FusedSequential(*stem(input_nc, ngf, norm))
            self.obj_downsampler = downs()
        if which_stream == 'ctx_label':
            self.feat_fuser = FeatureFusionBlock(self.feat_dim, feat_fusion.split('_')[1], norm)
        if 'early' in feat_fusion:
            self.latent_embedder = embedder(n_blocks)
        else:                                           # 'late': floor(n/2) blocks per stream, ceil(n/2) after the fusion
            self.obj_latent_embedder = embedder(n_blocks // 2)
            self.ctx_latent_embedder = embedder(n_blocks // 2)
            self.latent_embedder = embedder(n_blocks - n_blocks // 2)
        dec = []
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            dec += up(2 * c if (use_skip and i > 0) else c, c // 2, norm)
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
            if 'late' in self.feat_fusion:             # reference :212-214
                ctx = self.ctx_latent_embedder(ctx)
                obj = self.obj_latent_embedder(obj)
            h = self.feat_fuser(ctx, obj, m)           # fusion of ((1-m)*ctx, m*obj)
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
