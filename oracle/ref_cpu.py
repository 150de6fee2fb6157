"""CPU oracle for the mask2image training hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain PyTorch-fp32 (device='cpu') restatement of the reference algorithm for the
layout-to-image GAN training step.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package never does (it fails loudly without its HIP
library instead of falling back to this).

Parity pin: every class here is checked against the *imported* reference (``/root/reference`` through
``oracle/ref_shim.py``, which only works in the build container) by ``tests/golden/make_golden.py``;
the resulting arrays are committed under ``tests/golden/*.npz`` and re-checked by
``tests/test_oracle_golden.py`` on every run.  The one part that is **parity unpinned** is ``VGGLoss``
with the *real* ImageNet weights (torchvision ``vgg19(pretrained=True)``, un-vendored, 548 MB, no
network): it is pinned on seeded synthetic VGG weights instead.

All ``file:line`` citations are relative to the reference repository root.
"""
import math
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def _in_layer(ch):
    # models/layer_util.py:19-26  -> InstanceNorm2d(affine=False), eps 1e-5, biased variance
    return nn.InstanceNorm2d(ch, affine=False)


def _bn_layer(ch):
    # models/layer_util.py:20-21  -> BatchNorm2d(affine=True): weight / bias parameters + running statistics
    return nn.BatchNorm2d(ch, affine=True)


def get_norm_layer(norm_type='instance'):
    """models/layer_util.py:19-26 ('instance' | 'batch'; anything else raises there too)."""
    if norm_type == 'batch':
        return _bn_layer
    if norm_type == 'instance':
        return _in_layer
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


class ResnetBlock(nn.Module):
    """x + N(conv3(refpad(ReLU(N(conv3(refpad(x)))))))   (models/layer_util.py:333-378; no call site of the reference
    passes use_dropout=True)."""

    def __init__(self, dim, norm=_in_layer):
        super().__init__()
        # indices 1 and 5 carry the convolutions (2 and 6 the BatchNorm parameters), as in the reference Sequential
        self.conv_block = nn.Sequential(
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm(dim), nn.ReLU(False),
            nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm(dim))

    def forward(self, x):
        return x + self.conv_block(x)


def _stem(cin, ngf, norm=_in_layer):
    return [nn.ReflectionPad2d(3), nn.Conv2d(cin, ngf, 7), norm(ngf), nn.ReLU(False)]


def _down(c, norm=_in_layer):
    return [nn.Conv2d(c, 2 * c, 3, stride=2, padding=1), norm(2 * c), nn.ReLU(False)]


def _up(cin, cout, norm=_in_layer):
    return [nn.ConvTranspose2d(cin, cout, 3, stride=2, padding=1, output_padding=1),
            norm(cout), nn.ReLU(False)]


def _head(ngf, out_nc):
    return [nn.ReflectionPad2d(3), nn.Conv2d(ngf, out_nc, 7), nn.Tanh()]


class GlobalGenerator(nn.Module):
    """models/Pix2Pix_NET.py:63-101 (keys ``model.<i>.*``)."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, use_output_gate=False,
                 norm_layer='instance'):
        super().__init__()
        self.input_nc, self.output_nc, self.use_output_gate = input_nc, output_nc, use_output_gate
        norm = get_norm_layer(norm_layer)
        seq = _stem(input_nc, ngf, norm)
        for i in range(n_downsampling):
            seq += _down(ngf * 2 ** i, norm)
        seq += [ResnetBlock(ngf * 2 ** n_downsampling, norm) for _ in range(n_blocks)]
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            seq += _up(c, c // 2, norm)
        seq += _head(ngf, output_nc)
        self.model = nn.Sequential(*seq)

    def forward(self, x, mask=None):
        y = self.model(x)
        if self.use_output_gate and mask is not None:          # Pix2Pix_NET.py:96-99
            img = x[:, self.input_nc - 3:]
            y = (1 - mask) * img + mask * y
        return y


class LocalEnhancer(nn.Module):
    """models/Pix2Pix_NET.py:8-61 (defined in the reference, not reachable from its models)."""

    def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9,
                 n_local_enhancers=1, n_blocks_local=3, norm_layer='instance'):
        super().__init__()
        self.n_local_enhancers = n_local_enhancers
        norm = get_norm_layer(norm_layer)
        g = GlobalGenerator(input_nc, output_nc, ngf * 2 ** n_local_enhancers, n_downsample_global,
                            n_blocks_global, norm_layer=norm_layer).model
        self.model = nn.Sequential(*list(g.children())[:-3])                # :18
        for n in range(1, n_local_enhancers + 1):
            c = ngf * 2 ** (n_local_enhancers - n)
            down = _stem(input_nc, c, norm) + _down(c, norm)
            up = [ResnetBlock(2 * c, norm) for _ in range(n_blocks_local)] + _up(2 * c, c, norm)
            if n == n_local_enhancers:
                up += _head(ngf, output_nc)
            setattr(self, 'model%d_1' % n, nn.Sequential(*down))
            setattr(self, 'model%d_2' % n, nn.Sequential(*up))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)

    def forward(self, x):
        pyr = [x]
        for _ in range(self.n_local_enhancers):
            pyr.append(self.downsample(pyr[-1]))
        out = self.model(pyr[-1])
        for n in range(1, self.n_local_enhancers + 1):
            xi = pyr[self.n_local_enhancers - n]
            out = getattr(self, 'model%d_2' % n)(getattr(self, 'model%d_1' % n)(xi) + out)
        return out


class FeatureFusionBlock(nn.Module):
    """models/layer_util.py:295-330.  'add': x + y.  'concat': norm(conv1x1(ReLU(cat(x, y)))) with parameters at
    ``conv1`` (and ``norm1`` under --norm batch); ``main_module`` is an Identity at the one call site (Pix2Pix_NET.py:135)."""

    def __init__(self, planes, fusion_type, norm):
        super().__init__()
        assert fusion_type in ('add', 'concat')                            # :300
        self.fusion_type = fusion_type
        if fusion_type == 'concat':                                        # :310-314
            self.conv1 = nn.Conv2d(planes * 2, planes, kernel_size=1, stride=1, padding=0)
            self.norm1 = norm(planes)

    def forward(self, x, y):
        if self.fusion_type == 'add':
            return x + y
        return self.norm1(self.conv1(F.relu(torch.cat([x, y], 1))))          # :325-328


class GlobalTwoStreamGenerator(nn.Module):
    """models/Pix2Pix_NET.py:103-247: --feat_fusion early_add (the shipped mode) | early_concat | late_add | late_concat,
    --norm instance | batch."""

    def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, use_skip=False,
                 which_stream='ctx', use_output_gate=False, extra_embed=False, feat_fusion='early_add',
                 norm_layer='instance'):
        super().__init__()
        assert not ('label' not in which_stream and 'late' in feat_fusion)   # :108
        assert not ('ctx' not in which_stream and 'late' in feat_fusion)     # :109
        self.nd, self.use_skip, self.which_stream = n_downsampling, use_skip, which_stream
        self.use_output_gate, self.output_nc, self.feat_fusion = use_output_gate, output_nc, feat_fusion
        norm = get_norm_layer(norm_layer)
        feat = ngf * 2 ** n_downsampling
        self.feat_dim = feat

        def downs():
            seq = []
            for i in range(n_downsampling):
                seq += _down(ngf * 2 ** i, norm)
            return nn.Sequential(*seq)

        def embedder(n):                                                     # :166-174
            return nn.Sequential(*[ResnetBlock(feat, norm) for _ in range(n)])

        if 'ctx' in which_stream:
            self.ctx_inputEmbedder = nn.Sequential(*_stem(6 if extra_embed else 3, ngf, norm))
            self.ctx_downsampler = downs()
        if 'label' in which_stream:
            self.obj_inputEmbedder = nn.Sequential(*_stem(input_nc, ngf, norm))
            self.obj_downsampler = downs()
        if which_stream == 'ctx_label':                                      # :133-136
            self.feat_fuser = FeatureFusionBlock(feat, feat_fusion.split('_')[1], norm)
        if 'early' in feat_fusion:                                           # :137-144
            self.latent_embedder = embedder(n_blocks)
        elif 'late' in feat_fusion:
            self.obj_latent_embedder = embedder(n_blocks // 2)
            self.ctx_latent_embedder = embedder(n_blocks // 2)
            self.latent_embedder = embedder(n_blocks - n_blocks // 2)
        dec = []
        for i in range(n_downsampling):
            c = ngf * 2 ** (n_downsampling - i)
            dec += _up(2 * c if (use_skip and i > 0) else c, c // 2, norm)    # :183-184
        self.decoder = nn.Sequential(*dec)
        self.outputEmbedder = nn.Sequential(*_head(ngf, output_nc))

    def _encode(self, stem, down, x, want_skips):
        h, skips = stem(x), []
        for i, layer in enumerate(down):
            h = layer(h)
            if want_skips and i < self.nd * 3 - 1 and i % 3 == 2:            # :202
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
            m = F.max_pool2d(mask, 2 ** self.nd, 2 ** self.nd)              # :134,237-238
            if 'late' in self.feat_fusion:                                  # :212-214
                ctx = self.ctx_latent_embedder(ctx)
                obj = self.obj_latent_embedder(obj)
            h = self.feat_fuser((1 - m) * ctx, m * obj)                     # :215-217
        elif self.which_stream == 'ctx':
            h = ctx
        else:
            h = obj
        h = self.latent_embedder(h)
        for i, layer in enumerate(self.decoder):
            if self.use_skip and skips and i > 0 and i % 3 == 0:            # :223-224
                h = torch.cat((skips[-((i - 3) // 3) - 1], h), 1)
            h = layer(h)
        y = self.outputEmbedder(h)
        if self.use_output_gate:
            y = (1 - mask) * img[:, :3] + mask * y                          # :245
        return y


def _d_flat_first(n_layers):
    """models/Discriminator_NET.py:100-104: where block j starts inside the flattened ``NLayerDiscriminator.model``."""
    lens = [2] + [3] * n_layers + [1]
    return [sum(lens[:j]) for j in range(n_layers + 2)]


def _d_keys_out(module, sd, prefix, local_metadata):
    import re
    first, pat = _d_flat_first(module.n_layers), re.compile(re.escape(prefix) + r'scale(\d+)_layer(\d+)\.(\d+)\.(.*)$')
    items, meta = list(sd.items()), getattr(sd, '_metadata', None)
    sd.clear()
    for k, v in items:
        m = pat.match(k)
        sd['%slayer%s.%d.%s' % (prefix, m.group(1), first[int(m.group(2))] + int(m.group(3)), m.group(4)) if m else k] = v
    if meta is not None:
        sd._metadata = meta
    return sd


def _d_keys_in(module, sd, prefix, *unused):
    import re
    first, pat = _d_flat_first(module.n_layers), re.compile(re.escape(prefix) + r'layer(\d+)\.(\d+)\.(.*)$')
    for k in list(sd.keys()):
        m = pat.match(k)
        if m:
            n = int(m.group(2))
            j = max(jj for jj, f in enumerate(first) if f <= n)
            sd['%sscale%s_layer%d.%d.%s' % (prefix, m.group(1), j, n - first[j], m.group(3))] = sd.pop(k)


class MultiscaleDiscriminator(nn.Module):
    """models/Discriminator_NET.py:11-118.  getIntermFeat=True: keys ``scale<i>_layer<j>.0.*`` (:24-26); False
    (``--no_ganFeat_loss``, pix2pixHD_condImg_model.py:74-75): the reference registers each scale as one flattened
    Sequential ``layer<i>`` (:27-28, keys ``layer<i>.<n>.*``) -- the same modules on the same inputs, so the blocks stay
    separate here and only the state-dict names follow the reference."""

    def __init__(self, input_nc, ndf=64, n_layers=3, num_D=3, norm='instance', spectral_norm=False, getIntermFeat=True,
                 use_sigmoid=False):
        super().__init__()
        self.num_D, self.n_layers = num_D, n_layers
        if use_sigmoid and getIntermFeat:
            # models/Discriminator_NET.py:24-27 copies only model0 .. model<n_layers+1> of each NLayerDiscriminator: the
            # trailing nn.Sigmoid (:95-96) is dropped and nn.BCELoss (losses.py:19-20) is fed raw logits -- an error in torch
            raise ValueError('--no_lsgan needs --no_ganFeat_loss: with feature matching the reference drops the Sigmoid in '
                             'front of its BCELoss (models/Discriminator_NET.py:24-27)')
        if not getIntermFeat:
            self._register_state_dict_hook(_d_keys_out)
            self._register_load_state_dict_pre_hook(_d_keys_in, with_module=True)
        _in_layer = globals()['_in_layer'] if norm == 'instance' else nn.BatchNorm2d   # 'batch': box2mask D
        conv = globals()['SNConv2d'] if spectral_norm else nn.Conv2d   # the build's --sn_D wrap (not a reference flag)
        for i in range(num_D):
            blocks = [[conv(input_nc, ndf, 4, 2, 2), nn.LeakyReLU(0.2, False)]]
            nf = ndf
            for _ in range(1, n_layers):
                nf_prev, nf = nf, min(nf * 2, 512)
                blocks.append([conv(nf_prev, nf, 4, 2, 2), _in_layer(nf), nn.LeakyReLU(0.2, False)])
            nf_prev, nf = nf, min(nf * 2, 512)
            blocks.append([conv(nf_prev, nf, 4, 1, 2), _in_layer(nf), nn.LeakyReLU(0.2, False)])
            blocks.append([conv(nf, 1, 4, 1, 2)] + ([nn.Sigmoid()] if use_sigmoid else []))   # :95-96, inside ``layer<i>`` (:27-28)
            for j, b in enumerate(blocks):
                setattr(self, 'scale%d_layer%d' % (i, j), nn.Sequential(*b))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)

    def forward(self, x):
        out = []
        for i in range(self.num_D):
            feats, h = [], x
            for j in range(self.n_layers + 2):                               # :51 index reversal
                h = getattr(self, 'scale%d_layer%d' % (self.num_D - 1 - i, j))(h)
                feats.append(h)
            out.append(feats)
            if i != self.num_D - 1:
                x = self.downsample(x)
        return out


VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]


class Vgg19(nn.Module):
    """torchvision vgg19 ``features[0:30]`` (cfg 'E') cut into relu1_1 .. relu5_1
    (models/layer_util.py:380-411).  Keys ``slice<k>.<idx>.*`` with torchvision's feature indices."""
    SLICES = [(0, 2), (2, 7), (7, 12), (12, 21), (21, 30)]

    def __init__(self):
        super().__init__()
        layers, c = [], 3
        for v in VGG19_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(False)]
                c = v
        for k, (a, b) in enumerate(self.SLICES):
            seq = nn.Sequential()
            for idx in range(a, b):
                seq.add_module(str(idx), layers[idx])
            setattr(self, 'slice%d' % (k + 1), seq)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        out = []
        for k in range(5):
            x = getattr(self, 'slice%d' % (k + 1))(x)
            out.append(x)
        return out


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def gan_loss(pred_scales, target_is_real, use_lsgan=True):
    """LSGAN: sum over scales of mean((logits - t)^2) on the LAST tensor of each scale (models/losses.py:40-50);
    ``--no_lsgan``: nn.BCELoss on the Sigmoid outputs instead (losses.py:17-20)."""
    t = 1.0 if target_is_real else 0.0
    crit = F.mse_loss if use_lsgan else F.binary_cross_entropy
    return sum(crit(s[-1], torch.full_like(s[-1], t)) for s in pred_scales)


VGG_WEIGHTS = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]


def vgg_loss(vgg, x, y):
    """models/losses.py:75-82 (normalize=False default)."""
    fx, fy = vgg(x), vgg(y)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(VGG_WEIGHTS, fx, fy))


# ----------------------------------------------------------------------------------------------
# spectral norm (models/sn_utils.py:8-72)
# ----------------------------------------------------------------------------------------------
def l2normalize(v, eps=1e-12):
    return v / (torch.norm(v, p=2) + eps)


def max_singular_value(W, u, Ip=1):
    """Nothing is detached inside the iteration (models/sn_utils.py:11-25)."""
    W = W.reshape(W.size(0), -1)
    _u = u
    for _ in range(Ip):
        _v = l2normalize(_u @ W)
        _u = l2normalize(W @ _v.t()).view(1, -1)
    sigma = _u @ W @ _v.t()
    return sigma, _u


class SNLinear(nn.Linear):
    """models/sn_utils.py:28-47: y = x (W / sigma(W))^T + b; ``u`` moves on every training-mode forward.  (The reference
    re-registers u as a non-grad Parameter; a buffer under the same key here.)"""
    Ip = 1

    def __init__(self, in_features, out_features, bias=True):
        super().__init__(in_features, out_features, bias)
        self.register_buffer('u', torch.randn(1, out_features))

    @property
    def W_bar(self):
        sigma, _u = max_singular_value(self.weight, self.u, self.Ip)
        if self.training:
            self.u = _u.detach()
        return self.weight / sigma

    def forward(self, x):
        return F.linear(x, self.W_bar, self.bias)


class SNConv2d(nn.Conv2d):
    """models/sn_utils.py:49-72."""
    Ip = 1

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__(cin, cout, k, stride, padding, bias=bias)
        self.register_buffer('u', torch.randn(1, cout))

    @property
    def W_bar(self):
        sigma, _u = max_singular_value(self.weight, self.u, self.Ip)
        if self.training:
            self.u = _u.detach()
        return self.weight / sigma

    def forward(self, x):
        return F.conv2d(x, self.W_bar, self.bias, self.stride, self.padding)


# ----------------------------------------------------------------------------------------------
# the model (models/pix2pixHD_condImg_model.py, models/pix2pixHD_condImgColor_model.py)
# ----------------------------------------------------------------------------------------------
def get_edges(t):
    """models/pix2pixHD_condImg_model.py:285-291."""
    e = torch.zeros_like(t, dtype=torch.bool)
    dx = t[:, :, :, 1:] != t[:, :, :, :-1]
    dy = t[:, :, 1:, :] != t[:, :, :-1, :]
    e[:, :, :, 1:] |= dx
    e[:, :, :, :-1] |= dx
    e[:, :, 1:, :] |= dy
    e[:, :, :-1, :] |= dy
    return e.float()


def color_embedding(obj_mask, image, noise=None):
    """models/pix2pixHD_condImgColor_model.py:162-186; ``noise`` (B,3) replaces the U(0.97,1.03) draw."""
    s = (image * obj_mask).flatten(2).sum(2)
    cnt = obj_mask.flatten(1).sum(1, keepdim=True)
    emb = torch.where(cnt > 0, s / cnt.clamp(min=1e-30), torch.zeros_like(s))
    if noise is not None:
        emb = emb * noise
    return emb.clamp(-1, 1)


class Opt(dict):
    """Flag namespace with the reference defaults (options/mask2image_{base,train}_options.py)."""
    DEFAULTS = dict(
        model='pix2pixHD_condImg', netG='global', ngf=64, n_downsample_global=4, n_blocks_global=9,
        n_blocks_local=3, n_local_enhancers=1, norm='instance', label_nc=35, output_nc=3,
        no_instance=False, which_encoder='ctx', use_output_gate=False, use_skip=False,
        feat_fusion='early_add', num_D=2, n_layers_D=3, ndf=64, lambda_feat=10.0, lambda_rec=0.0,
        no_ganFeat_loss=False, no_vgg_loss=False, no_lsgan=False, pool_size=0, no_imgCond=False,
        mask_gan_input=False, use_soft_mask=False, lr=2e-4, beta1=0.5, niter=100, niter_decay=100,
        isTrain=True, batchSize=1, sn_D=False)

    def __init__(self, **kw):
        super().__init__(self.DEFAULTS)
        self.update(kw)

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class Mask2ImageModel(nn.Module):
    """Pix2PixHDModel_condImg / _condImgColor: nets, 5 losses, two Adams
    (models/pix2pixHD_condImg_model.py:24-259; colour variant pix2pixHD_condImgColor_model.py:147-228)."""
    loss_names = ['G_GAN', 'G_GAN_Feat', 'G_VGG', 'D_real', 'D_fake']

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.color = opt.model == 'pix2pixHD_condImgColor'
        nc = opt.label_nc + (0 if opt.no_instance else 1)
        if opt.netG == 'global':
            self.netG = GlobalGenerator(nc + 3, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                        opt.n_blocks_global, opt.use_output_gate, norm_layer=opt.norm)
        elif opt.netG == 'global_twostream':
            self.netG = GlobalTwoStreamGenerator(nc, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                                 opt.n_blocks_global, opt.use_skip, opt.which_encoder,
                                                 opt.use_output_gate, extra_embed=self.color,
                                                 feat_fusion=opt.feat_fusion, norm_layer=opt.norm)
        elif opt.netG == 'local':
            self.netG = LocalEnhancer(nc + 3, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                      opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local,
                                      norm_layer=opt.norm)
        else:
            raise NameError('global generator name is not defined properly: %s' % opt.netG)
        d_in = nc + opt.output_nc + (0 if opt.no_imgCond else 3)             # :66-71
        if opt.netG == 'global_twostream' and opt.which_encoder == 'ctx':
            d_in = 3
        self.d_in = d_in
        self.netD = MultiscaleDiscriminator(d_in, opt.ndf, opt.n_layers_D, opt.num_D, norm=opt.norm, spectral_norm=opt.sn_D,
                                            getIntermFeat=not opt.no_ganFeat_loss, use_sigmoid=opt.no_lsgan)
        self.vgg = None if opt.no_vgg_loss else Vgg19()
        self.optimizer_G = torch.optim.Adam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
        self.optimizer_D = torch.optim.Adam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))

    # --- input encoding (:144-174) ---
    def encode_input(self, label, inst, image, mask_in, obj_mask=None, color_noise=None):
        o = self.opt
        if o.label_nc == 0:
            onehot = label
        else:
            onehot = torch.zeros(label.size(0), o.label_nc, label.size(2), label.size(3))
            onehot.scatter_(1, label.long(), 1.0)
        if not o.no_instance:
            onehot = torch.cat((onehot, get_edges(inst)), 1)
        cond = (1 - mask_in) * image
        if self.color:
            emb = color_embedding(obj_mask, image, color_noise)
            cond = torch.cat((cond, emb.view(emb.size(0), 3, 1, 1) * mask_in), 1)
        return onehot, cond

    def _d_input(self, cond, img, mask):
        o = self.opt
        x = img if (o.netG == 'global_twostream' and o.which_encoder == 'ctx') else torch.cat((cond, img), 1)
        return x * mask if o.mask_gan_input else x

    def generate(self, onehot, cond, mask_in):
        if self.opt.netG == 'global':
            return self.netG(torch.cat((onehot, cond), 1), mask_in)
        if self.opt.netG == 'local':
            return self.netG(torch.cat((onehot, cond), 1))
        return self.netG(cond, onehot, mask_in)

    # --- the training graph (:198-259) ---
    def forward(self, label, inst, image, feat, mask_in, mask_out, obj_mask=None, infer=False,
                color_noise=None):
        o = self.opt
        onehot, cond = self.encode_input(label, inst, image, mask_in, obj_mask, color_noise)
        fake = self.generate(onehot, cond, mask_in)
        d_cond = onehot if o.no_imgCond else torch.cat((onehot, cond), 1)
        m = mask_out if o.use_soft_mask else mask_in
        pred_fake_pool = self.netD(self._d_input(d_cond, fake.detach(), m))
        loss_D_fake = gan_loss(pred_fake_pool, False, not o.no_lsgan)
        pred_real = self.netD(self._d_input(d_cond, image, m))
        loss_D_real = gan_loss(pred_real, True, not o.no_lsgan)
        pred_fake = self.netD(self._d_input(d_cond, fake, m))
        loss_G_GAN = gan_loss(pred_fake, True, not o.no_lsgan)
        loss_feat = torch.zeros(1)
        if not o.no_ganFeat_loss:
            feat_w, d_w = 4.0 / (o.n_layers_D + 1), 1.0 / o.num_D
            for i in range(o.num_D):
                for j in range(len(pred_fake[i]) - 1):
                    # evaluation order as in :240-242: ((D_w*feat_w) * L1) * lambda_feat, in fp32
                    loss_feat = loss_feat + d_w * feat_w * F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) \
                        * o.lambda_feat
        loss_vgg = torch.zeros(1)
        if self.vgg is not None:
            loss_vgg = vgg_loss(self.vgg, fake, image) * o.lambda_feat
        if o.lambda_rec > 0:
            loss_feat = loss_feat + F.l1_loss(fake, image) * o.lambda_rec
        return [[loss_G_GAN, loss_feat, loss_vgg, loss_D_real, loss_D_fake], fake if infer else None]

    # --- one optimisation step: train_mask2image.py:58-86 ---
    def optimize_parameters(self, batch, **kw):
        losses, _ = self.forward(batch['label'], batch['inst'], batch['image'], None, batch['mask_in'],
                                 batch['mask_out'], batch.get('obj_mask'), **kw)
        ld = dict(zip(self.loss_names, [x.mean() for x in losses]))
        loss_D = (ld['D_fake'] + ld['D_real']) * 0.5
        loss_G = ld['G_GAN'] + ld['G_GAN_Feat'] + ld['G_VGG']
        self.optimizer_G.zero_grad()
        loss_G.backward()
        self.optimizer_G.step()
        self.optimizer_D.zero_grad()
        loss_D.backward()
        self.optimizer_D.step()
        return OrderedDict((k, float(v.detach())) for k, v in ld.items())
