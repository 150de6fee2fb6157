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
sigmoid heads.  ``norm_layer`` selects BatchNorm2d ('batch', scripts/train_box2mask_city.sh) or InstanceNorm2d(affine=False)
('instance', scripts/train_box2mask_ade.sh); ``add_dilated_layers`` (ADE) prepends two DilatedResnetBlocks (dilation 2, 4)
to the latent encoder."""
import math

import torch.nn as nn

from .. import ops
from .. import nn as hn
from ..nn import Conv2d, ConvTranspose2d, BatchNorm2d, InstanceNorm2d, ReflectionPad2d, ReLU, _pw


class Upsample(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear') marker (no parameters)."""

    def __init__(self, align_corners=False):
        super().__init__()
        self.align_corners = align_corners


def _norm_factory(kind):
    """get_norm_layer (reference models/layer_util.py:19-26)."""
    if kind == 'batch':
        return BatchNorm2d
    if kind == 'instance':
        return InstanceNorm2d
    raise NotImplementedError('normalization layer [%s] is not found' % kind)


def _conv(l, x, pad_mode='zero', pad=None, norm=None):
    """``norm``: the normalisation applied right behind this conv; its mean subtraction makes the bias gradient exactly
    zero (see nn.run_layers), so the bias-gradient pass is skipped."""
    b = _pw(l.bias)
    if b is not None and hn.SCHED.dead_bias_skip and (isinstance(norm, InstanceNorm2d) or
                                                 (isinstance(norm, BatchNorm2d) and norm.training)):
        l.bias._him_dead_grad = True
        b = b.detach()
    elif b is not None and getattr(l.bias, '_him_dead_grad', False):
        l.bias._him_dead_grad = False       # BatchNorm went to eval(): the bias gradient is live again (nn.run_layers)
    return ops.conv2d(x, _pw(l.weight), b, l.stride, l.padding if pad is None else pad, pad_mode, 'none', 0.0)


class ConvResnetBlock(nn.Module):
    """act(x) -> [conv k s] -> BN  +  shortcut(act(x)) = conv1x1 s -> BN   (layer_util.py:128-171, num_layers 1);
    ``activation``: ReLU in the generator, LeakyReLU(0.2) in NLayerResDiscriminator (Discriminator_NET.py:136-152)."""

    def __init__(self, cin, cout, stride, k, norm=BatchNorm2d, activation=None):
        super().__init__()
        self.shortcut = None if (cin == cout and stride == 1) else nn.Sequential(Conv2d(cin, cout, 1, stride, 0),
                                                                                 norm(cout))
        self.deep = nn.Sequential(activation or ReLU(), Conv2d(cin, cout, k, stride, (k - 1) // 2), norm(cout))

    def forward(self, x):
        r = ops.activation(x, self.deep[0].act, getattr(self.deep[0], 'slope', 0.0))
        res = r if self.shortcut is None else self.shortcut[1].apply_to(_conv(self.shortcut[0], r, norm=self.shortcut[1]))
        out = self.deep[2].apply_to(_conv(self.deep[1], r, norm=self.deep[2]), residual=res)
        return out, r


class DeconvResnetBlock(nn.Module):
    """relu(x) -> ConvTranspose2d(k4, s2, p1) -> BN  +  shortcut(relu(x)) = [conv1x1 -> BN] -> bilinear x2
    (layer_util.py:173-250, even kernel, num_layers 1)."""

    def __init__(self, cin, cout, stride, k, align_corners, norm=BatchNorm2d):
        super().__init__()
        if k % 2 or stride != 2:
            raise NotImplementedError('DeconvResnetBlock: only the even-kernel stride-2 (ConvTranspose2d) form is on the HIP path')
        sc = []
        if cin != cout:
            sc += [Conv2d(cin, cout, 1, 1, 0), norm(cout)]
        sc += [Upsample(align_corners)]
        self.shortcut = nn.Sequential(*sc)
        self.deep = nn.Sequential(ReLU(), ConvTranspose2d(cin, cout, k, stride, (k - 1) // 2, stride - 2), norm(cout))

    def forward(self, x):
        r = ops.activation(x, 'relu')
        res = r
        if len(self.shortcut) == 3:
            res = self.shortcut[1].apply_to(_conv(self.shortcut[0], res, norm=self.shortcut[1]))
        res = ops.upsample_bilinear2(res, self.shortcut[-1].align_corners)
        d = self.deep[1]
        b, nrm = _pw(d.bias), self.deep[2]
        if b is not None and hn.SCHED.dead_bias_skip and (isinstance(nrm, InstanceNorm2d) or nrm.training):
            d.bias._him_dead_grad = True
            b = b.detach()
        elif b is not None and getattr(d.bias, '_him_dead_grad', False):
            d.bias._him_dead_grad = False
        y = ops.conv_transpose2d(r, _pw(d.weight), b, d.stride, d.padding, d.output_padding, 'none', 0.0)
        return nrm.apply_to(y, residual=res)


class BNResnetBlock(nn.Module):
    """x + BN(conv3(refpad(ReLU(BN(conv3(refpad(x)))))))   (layer_util.py:333-378 with norm_layer = BatchNorm2d)."""

    def __init__(self, dim, norm=BatchNorm2d):
        super().__init__()
        self.conv_block = nn.Sequential(ReflectionPad2d(1), Conv2d(dim, dim, 3), norm(dim), ReLU(),
                                        ReflectionPad2d(1), Conv2d(dim, dim, 3), norm(dim))

    def forward(self, x):
        cb = self.conv_block
        h = cb[2].apply_to(_conv(cb[1], x, 'reflect', 1, cb[2]), 'relu')
        return cb[6].apply_to(_conv(cb[5], h, 'reflect', 1, cb[6]), residual=x)


class _BiasFreeConv3x3(nn.Module):
    """conv3x3(..., bias=False, dilation=d, padding=d) parameter holder (reference models/layer_util.py:254-256)."""

    def __init__(self, cin, cout, dilation):
        super().__init__()
        import torch
        import math
        self.dilation = dilation
        # construction-time init of the nn.Conv2d the reference builds here (kaiming_uniform, a = sqrt(5)):
        # U(+-1/sqrt(fan_in)); the box2mask generator is never passed through weights_init
        bound = 1.0 / math.sqrt(cin * 9)
        self.weight = nn.Parameter((torch.rand(cout, cin, 3, 3) * 2 - 1) * bound)


class DilatedResnetBlock(nn.Module):
    """relu(norm(conv2(relu(norm(conv1(x))))) + x) with bias-free dilated 3x3 convs (reference layer_util.py:259-293;
    the ReLU comes AFTER the residual add).  State-dict keys conv1.weight / conv2.weight (+ bn1 / bn2 for BatchNorm).
    Each dilated conv runs as the plain pad-1 conv on the dilation^2 phase images (ops.dilated_conv3x3)."""

    def __init__(self, dim, dilation, norm):
        super().__init__()
        self.conv1 = _BiasFreeConv3x3(dim, dim, dilation[0])
        self.bn1 = norm(dim)
        self.relu = ReLU()
        self.conv2 = _BiasFreeConv3x3(dim, dim, dilation[1])
        self.bn2 = norm(dim)

    def forward(self, x):
        h = self.bn1.apply_to(ops.dilated_conv3x3(x, _pw(self.conv1.weight), self.conv1.dilation), 'relu')
        h = self.bn2.apply_to(ops.dilated_conv3x3(h, _pw(self.conv2.weight), self.conv2.dilation), residual=x)
        return ops.activation(h, 'relu')


class _SimpleResBlock(nn.Module):
    """Shared body of --use_simpleRes' two stages (reference MaskTwoStreamConv*_NET.py:228-306): main path conv3 -> norm ->
    ReLU -> conv, side path one conv of the same input, then norm -> ReLU of their sum.  State-dict keys
    ``main_path.{0,1,3}``, ``side_path``, ``output_layer.0`` as upstream."""

    def __init__(self, cin, cout, norm, k2, s2, p2, ks, ss, ps):
        super().__init__()
        self.main_path = nn.Sequential(Conv2d(cin, cin, 3, 1, 1), norm(cin), ReLU(), Conv2d(cin, cout, k2, s2, p2))
        self.side_path = Conv2d(cin, cout, ks, ss, ps)
        self.output_layer = nn.Sequential(norm(cout), ReLU())

    def _body(self, x):
        m = self.main_path
        h = m[1].apply_to(_conv(m[0], x, norm=m[1]), 'relu')
        # both biases feed the sum that ``output_layer``'s norm centres: together they shift the plane mean only
        s = ops.add(_conv(self.side_path, x, norm=self.output_layer[0]), _conv(m[3], h, norm=self.output_layer[0]))
        return self.output_layer[0].apply_to(s, 'relu')


class DownResBlock3x3(_SimpleResBlock):
    """``downResBlock_3x3``: main ... -> conv4 s2 p1, side conv4 s2 p1.  Returns (output, input): nothing rectifies the
    input in place here, the decoder's skip feature is the input itself."""

    def __init__(self, cin, cout, norm=BatchNorm2d):
        super().__init__(cin, cout, norm, 4, 2, 1, 4, 2, 1)

    def forward(self, x):
        return self._body(x), x


class UpResBlock3x3(_SimpleResBlock):
    """``upResBlock_3x3``: bilinear x2 (align_corners False), main ... -> conv3 p1, side conv1x1."""

    def __init__(self, cin, cout, norm=BatchNorm2d):
        super().__init__(cin, cout, norm, 3, 1, 1, 1, 1, 0)

    def forward(self, x):
        return self._body(ops.upsample_bilinear2(x, False))


class MaskTwoStreamConvSwitch_NET(nn.Module):
    """``comb`` False: MaskTwoStreamConvSwitch_NET (--no_comb, every shipped recipe).  True: the parser's default
    MaskTwoStreamConv_NET -- the same modules (no dilated blocks), returning the object-gated combination of the two
    streams' logits (MaskTwoStreamConv_NET.py:206-223)."""
    comb = False

    def __init__(self, opt):
        super().__init__()
        g = lambda k, d: getattr(opt, k, d)  # noqa: E731
        norm = _norm_factory(g('norm_layer', 'batch'))
        simple = bool(g('use_simpleRes', False))
        self.which_stream = g('which_stream', 'obj_context')
        if 'obj' not in self.which_stream and 'context' not in self.which_stream:
            raise AssertionError('which_stream [%s] holds neither obj nor context' % self.which_stream)   # reference :18
        self.num_layers = g('num_layers', 3)
        label_nc = g('label_nc', 35)
        input_nc = label_nc * 2 if g('cond_in', 'ctx_obj') == 'ctx_obj' else label_nc
        k, n_blocks = g('conv_size', 4), g('n_blocks', 6)
        align = bool(g('upsample_align_corners', False))
        dims = [g('conv_dim', 64), 96, 128, 256, 512]      # hard-coded in the reference (:26)
        enc = [Conv2d(input_nc, dims[0], 7, 2, 3), norm(dims[0]), ReLU()]
        for i in range(self.num_layers):
            enc.append(DownResBlock3x3(dims[i], dims[i + 1], norm) if simple else
                       ConvResnetBlock(dims[i], dims[i + 1], 2, k, norm))
        self.conv_encoder_modules = nn.Sequential(*enc)
        latent = dims[self.num_layers]
        lat = []
        if g('add_dilated_layers', False) and not self.comb:   # the ADE recipe (MaskTwoStreamConvSwitch_NET.py:103-105)
            lat += [DilatedResnetBlock(latent, (2, 2), norm), DilatedResnetBlock(latent, (4, 4), norm)]
        lat += [BNResnetBlock(latent, norm) for _ in range(int(math.floor(n_blocks / 2)))]
        self.latent_encoder = nn.Sequential(*lat)

        def decoder(out_nc, skip):
            layers, od = [], latent
            for i in range(self.num_layers + 1):
                idim = od
                od = dims[self.num_layers - i - 1] if i < self.num_layers else idim // 2
                if skip and 1 <= i <= self.num_layers:
                    idim *= 2
                layers.append(UpResBlock3x3(idim, od, norm) if simple else DeconvResnetBlock(idim, od, 2, k, align, norm))
            layers.append(Conv2d(od, out_nc, 3, 1, 1))
            return nn.Sequential(*layers)

        def latent_dec():
            return nn.Sequential(*[BNResnetBlock(latent, norm) for _ in range(int(math.ceil(n_blocks / 2)))])

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
        f = e[1].apply_to(_conv(e[0], input_var, norm=e[1]), 'relu')
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
        if not self.comb:
            return ctx_logit, ctx_prob, obj_logit, obj_prob
        comb_logit, comb_prob = ctx_logit, ctx_prob
        if ctx_logit is not None and obj_logit is not None:
            comb_logit = ops.gate_comb(ctx_logit, obj_prob, obj_logit)       # (1 - p) * ctx + p * obj
            comb_prob = ops.log_softmax_channels(comb_logit)
        return comb_logit, comb_prob, obj_logit, obj_prob


class MaskTwoStreamConv_NET(MaskTwoStreamConvSwitch_NET):
    """reference models/MaskTwoStreamConv_NET.py (what TwoStreamAE_mask builds without --no_comb)."""
    comb = True
