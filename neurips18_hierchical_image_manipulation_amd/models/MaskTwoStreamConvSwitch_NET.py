"""box2mask generator on the HIP kernels: the reference's ``MaskTwoStreamConvSwitch_NET``
(models/MaskTwoStreamConvSwitch_NET.py:13-208) with ConvResnetBlock / DeconvResnetBlock / ResnetBlock of
models/layer_util.py:128-250,333-378.  Module tree, state_dict keys and shapes are the reference's
(``conv_encoder_modules.3.deep.1.weight`` ...), so checkpoints interchange.

The reference applies ``nn.ReLU(True)`` IN PLACE to every block input (layer_util.py:150,196-205), which also rectifies
the tensor the shortcut path reads and the encoder features kept for the skip connections
(MaskTwoStreamConvSwitch_NET.py:164-167,179-183); here that is one out-of-place activation kernel per block whose
result feeds both paths and is handed to the decoder as the skip feature (oracle/ref_mask_cpu.py restates the same).

Kernels per block: conv / transposed-conv (MFMA implicit GEMM), BatchNorm2d (batch statistics in training mode,
running statistics in eval mode) fused with the residual add, bilinear x2 of the shortcut, channel log-softmax /
sigmoid heads.  BatchNorm / 'instance' selection follows ``norm_layer`` ('batch' is what scripts/train_box2mask_city.sh
trains)."""
import math

import torch.nn as nn

from .. import ops
from ..nn import Conv2d, ConvTranspose2d, BatchNorm2d, ReflectionPad2d, ReLU, _pw


class Upsample(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear') marker (no parameters)."""

    def __init__(self, align_corners=False):
        super().__init__()
        self.align_corners = align_corners


def _conv(l, x, pad_mode='zero', pad=None):
    return ops.conv2d(x, _pw(l.weight), _pw(l.bias), l.stride, l.padding if pad is None else pad, pad_mode, 'none', 0.0)


class ConvResnetBlock(nn.Module):
    """relu(x) -> [conv k s] -> BN  +  shortcut(relu(x)) = conv1x1 s -> BN   (layer_util.py:128-171, num_layers 1)."""

    def __init__(self, cin, cout, stride, k):
        super().__init__()
        self.shortcut = None if (cin == cout and stride == 1) else nn.Sequential(Conv2d(cin, cout, 1, stride, 0),
                                                                                 BatchNorm2d(cout))
        self.deep = nn.Sequential(ReLU(), Conv2d(cin, cout, k, stride, (k - 1) // 2), BatchNorm2d(cout))

    def forward(self, x):
        r = ops.activation(x, 'relu')
        res = r if self.shortcut is None else self.shortcut[1].apply_to(_conv(self.shortcut[0], r))
        out = self.deep[2].apply_to(_conv(self.deep[1], r), residual=res)
        return out, r


class DeconvResnetBlock(nn.Module):
    """relu(x) -> ConvTranspose2d(k4, s2, p1) -> BN  +  shortcut(relu(x)) = [conv1x1 -> BN] -> bilinear x2
    (layer_util.py:173-250, even kernel, num_layers 1)."""

    def __init__(self, cin, cout, stride, k, align_corners):
        super().__init__()
        if k % 2 or stride != 2:
            raise NotImplementedError('DeconvResnetBlock: only the even-kernel stride-2 (ConvTranspose2d) form is on the HIP path')
        sc = []
        if cin != cout:
            sc += [Conv2d(cin, cout, 1, 1, 0), BatchNorm2d(cout)]
        sc += [Upsample(align_corners)]
        self.shortcut = nn.Sequential(*sc)
        self.deep = nn.Sequential(ReLU(), ConvTranspose2d(cin, cout, k, stride, (k - 1) // 2, stride - 2), BatchNorm2d(cout))

    def forward(self, x):
        r = ops.activation(x, 'relu')
        res = r
        if len(self.shortcut) == 3:
            res = self.shortcut[1].apply_to(_conv(self.shortcut[0], res))
        res = ops.upsample_bilinear2(res, self.shortcut[-1].align_corners)
        d = self.deep[1]
        y = ops.conv_transpose2d(r, _pw(d.weight), _pw(d.bias), d.stride, d.padding, d.output_padding, 'none', 0.0)
        return self.deep[2].apply_to(y, residual=res)


class BNResnetBlock(nn.Module):
    """x + BN(conv3(refpad(ReLU(BN(conv3(refpad(x)))))))   (layer_util.py:333-378 with norm_layer = BatchNorm2d)."""

    def __init__(self, dim):
        super().__init__()
        self.conv_block = nn.Sequential(ReflectionPad2d(1), Conv2d(dim, dim, 3), BatchNorm2d(dim), ReLU(),
                                        ReflectionPad2d(1), Conv2d(dim, dim, 3), BatchNorm2d(dim))

    def forward(self, x):
        cb = self.conv_block
        h = cb[2].apply_to(_conv(cb[1], x, 'reflect', 1), 'relu')
        return cb[6].apply_to(_conv(cb[5], h, 'reflect', 1), residual=x)


class MaskTwoStreamConvSwitch_NET(nn.Module):
    def __init__(self, opt):
        super().__init__()
        g = lambda k, d: getattr(opt, k, d)  # noqa: E731
        if g('norm_layer', 'batch') != 'batch':
            raise NotImplementedError('box2mask generator: only norm_layer=batch is on the HIP path')
        if g('use_simpleRes', False) or g('add_dilated_layers', False):
            raise NotImplementedError('use_simpleRes / add_dilated_layers are not on the HIP path')
        self.which_stream = g('which_stream', 'obj_context')
        self.num_layers = g('num_layers', 3)
        label_nc = g('label_nc', 35)
        input_nc = label_nc * 2 if g('cond_in', 'ctx_obj') == 'ctx_obj' else label_nc
        k, n_blocks = g('conv_size', 4), g('n_blocks', 6)
        align = bool(g('upsample_align_corners', False))
        dims = [g('conv_dim', 64), 96, 128, 256, 512]      # hard-coded in the reference (:26)
        enc = [Conv2d(input_nc, dims[0], 7, 2, 3), BatchNorm2d(dims[0]), ReLU()]
        for i in range(self.num_layers):
            enc.append(ConvResnetBlock(dims[i], dims[i + 1], 2, k))
        self.conv_encoder_modules = nn.Sequential(*enc)
        latent = dims[self.num_layers]
        self.latent_encoder = nn.Sequential(*[BNResnetBlock(latent) for _ in range(int(math.floor(n_blocks / 2)))])

        def decoder(out_nc, skip):
            layers, od = [], latent
            for i in range(self.num_layers + 1):
                idim = od
                od = dims[self.num_layers - i - 1] if i < self.num_layers else idim // 2
                if skip and 1 <= i <= self.num_layers:
                    idim *= 2
                layers.append(DeconvResnetBlock(idim, od, 2, k, align))
            layers.append(Conv2d(od, out_nc, 3, 1, 1))
            return nn.Sequential(*layers)

        def latent_dec():
            return nn.Sequential(*[BNResnetBlock(latent) for _ in range(int(math.ceil(n_blocks / 2)))])

        if 'obj' in self.which_stream:
            self.obj_conv_decoder_modules = decoder(1, False)
            self.obj_latent_decoder = latent_dec()
        if 'context' in self.which_stream:
            self.ctx_conv_decoder_modules = decoder(g('output_nc', 35), True)
            self.ctx_latent_decoder = latent_dec()

    def initialize(self):
        """reference API (MaskContextAE_NET.initialize builds the modules; they already exist here)."""
        return self

    @property
    def trainable_parameters(self):
        return list(self.parameters())

    def _decode(self, dec, feat, skips):
        n = len(dec)
        for i in range(n - 1):
            if skips is not None and 1 <= i <= self.num_layers:
                feat = ops.cat_channels([skips[-i], feat], None, 1)
            feat = dec[i](feat)
        return feat

    def forward(self, input_var, cls_onehot=None, is_bkg=False):
        e = self.conv_encoder_modules
        f = e[1].apply_to(_conv(e[0], input_var), 'relu')
        skips = []
        for i in range(3, 3 + self.num_layers):
            f, r = e[i](f)
            skips.append(r)
        for blk in self.latent_encoder:
            f = blk(f)
        ctx_logit = ctx_prob = obj_logit = obj_prob = None
        if 'context' in self.which_stream:
            h = f
            for blk in self.ctx_latent_decoder:
                h = blk(h)
            dec = self.ctx_conv_decoder_modules
            ctx_logit = _conv(dec[-1], self._decode(dec, h, skips))
            ctx_prob = ops.log_softmax_channels(ctx_logit)
        if 'obj' in self.which_stream:
            h = f
            for blk in self.obj_latent_decoder:
                h = blk(h)
            dec = self.obj_conv_decoder_modules
            obj_logit = _conv(dec[-1], self._decode(dec, h, None))
            obj_prob = ops.activation(obj_logit, 'sigmoid')
        return ctx_logit, ctx_prob, obj_logit, obj_prob
