"""CPU restatement of the reference's box2mask generator  --  TEST INFRASTRUCTURE (see oracle/ref_cpu.py header).

MaskTwoStreamConvSwitch_NET (reference models/MaskTwoStreamConvSwitch_NET.py:13-208) with the blocks of
models/layer_util.py: ConvResnetBlock (:128-171), DeconvResnetBlock (:173-250), ResnetBlock (:333-378).  State-dict keys
and shapes are the reference's.  Pinned against the imported reference class by tests/golden/make_golden.py (fixture
box2mask_net.npz: forward in training and eval mode; parameter gradients of the reference run under
``torch.autograd.graph.allow_mutation_on_saved_tensors`` -- its in-place ReLUs alias saved tensors, which torch >= 1.x
rejects otherwise).

Aliasing quirk restated OUT OF PLACE.  Every (De)ConvResnetBlock starts its deep path with ``nn.ReLU(True)`` applied
to the block input itself (layer_util.py:150-153, 196-205): the tensor is rectified IN PLACE, so
  * the shortcut path reads relu(x), not x (layer_util.py:166-170: ``residual = x`` is the same tensor), and
  * the encoder features kept for the skip connections (MaskTwoStreamConvSwitch_NET.py:179-183) have been rectified by
    the NEXT encoder block by the time the decoder concatenates them (:164-167).
nn.Upsample(mode='bilinear') is evaluated with align_corners=False (what the reference code does under the torch of
this container; torch 0.3.1 used align_corners=True -- flag ``align_corners``).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Up(nn.Module):
    def __init__(self, align):
        super().__init__()
        self.align = align

    def forward(self, x):
        return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=self.align)


def _norm(kind):
    """get_norm_layer (layer_util.py:19-26): 'batch' -> BatchNorm2d(affine=True), 'instance' -> InstanceNorm2d(affine=False)"""
    if kind == 'batch':
        return nn.BatchNorm2d
    if kind == 'instance':
        return lambda ch: nn.InstanceNorm2d(ch, affine=False)
    raise NotImplementedError(kind)


class ConvResnetBlock(nn.Module):   # layer_util.py:128-171 (num_layers = 1)
    def __init__(self, cin, cout, stride, k, norm=nn.BatchNorm2d, slope=0.0):
        super().__init__()
        self.shortcut = None if (cin == cout and stride == 1) else nn.Sequential(nn.Conv2d(cin, cout, 1, stride),
                                                                                 norm(cout))
        # activation_fn: nn.ReLU(True) in the generator, nn.LeakyReLU(0.2, True) in NLayerResDiscriminator
        self.deep = nn.Sequential(nn.LeakyReLU(slope) if slope else nn.ReLU(), nn.Conv2d(cin, cout, k, stride, (k - 1) // 2),
                                  norm(cout))

    def forward(self, x):
        r = self.deep[0](x)                # the in-place activation: both paths (and the caller's alias) see act(x)
        res = r if self.shortcut is None else self.shortcut(r)
        return self.deep[2](self.deep[1](r)) + res, r


class DeconvResnetBlock(nn.Module):  # layer_util.py:173-250 (even kernel -> ConvTranspose2d, num_layers = 1)
    def __init__(self, cin, cout, stride, k, align, norm=nn.BatchNorm2d):
        super().__init__()
        sc = []
        if cin != cout:
            sc += [nn.Conv2d(cin, cout, 1), norm(cout)]
        if stride > 1:
            sc += [_Up(align)]
        self.shortcut = nn.Sequential(*sc) if sc else None
        assert k % 2 == 0 and stride > 1
        self.deep = nn.Sequential(nn.ReLU(), nn.ConvTranspose2d(cin, cout, k, stride, (k - 1) // 2, stride - 2),
                                  norm(cout))

    def forward(self, x):
        r = F.relu(x)
        res = r if self.shortcut is None else self.shortcut(r)
        return self.deep[2](self.deep[1](r)) + res


class ResnetBlock(nn.Module):        # layer_util.py:333-378, reflect padding, BatchNorm / InstanceNorm
    def __init__(self, dim, norm=nn.BatchNorm2d):
        super().__init__()
        self.conv_block = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm(dim), nn.ReLU(),
                                        nn.ReflectionPad2d(1), nn.Conv2d(dim, dim, 3), norm(dim))

    def forward(self, x):
        return x + self.conv_block(x)


class DilatedResnetBlock(nn.Module):  # layer_util.py:259-293: bias-free dilated conv3x3 pair, ReLU AFTER the residual add
    def __init__(self, dim, dilation, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, dim, 3, 1, dilation, dilation, bias=False)
        self.bn1 = norm(dim)
        self.relu = nn.ReLU()
        self.conv2 = nn.Conv2d(dim, dim, 3, 1, dilation, dilation, bias=False)
        self.bn2 = norm(dim)

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + x)


class DownResBlock3x3(nn.Module):
    """--use_simpleRes encoder stage (``downResBlock_3x3``, MaskTwoStreamConv*_NET.py:270-306): conv3 -> norm -> ReLU ->
    conv4 s2 p1, plus the side path conv4 s2 p1 of the input, then norm -> ReLU.  Returns (output, input) like
    ConvResnetBlock above: nothing modifies the input in place here, the decoder's skip is the input as it is."""

    def __init__(self, cin, cout, norm):
        super().__init__()
        self.main_path = nn.Sequential(nn.Conv2d(cin, cin, 3, 1, 1), norm(cin), nn.ReLU(), nn.Conv2d(cin, cout, 4, 2, 1))
        self.side_path = nn.Conv2d(cin, cout, 4, 2, 1)
        self.output_layer = nn.Sequential(norm(cout), nn.ReLU())

    def forward(self, x):
        return self.output_layer(self.side_path(x) + self.main_path(x)), x


class UpResBlock3x3(nn.Module):
    """--use_simpleRes decoder stage (``upResBlock_3x3``, :228-268): bilinear x2 (F.upsample's default: align_corners
    False), conv3 -> norm -> ReLU -> conv3 plus the side path conv1x1, then norm -> ReLU."""

    def __init__(self, cin, cout, norm):
        super().__init__()
        self.main_path = nn.Sequential(nn.Conv2d(cin, cin, 3, 1, 1), norm(cin), nn.ReLU(), nn.Conv2d(cin, cout, 3, 1, 1))
        self.side_path = nn.Conv2d(cin, cout, 1, 1, 0)
        self.output_layer = nn.Sequential(norm(cout), nn.ReLU())

    def forward(self, x):
        x = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        return self.output_layer(self.side_path(x) + self.main_path(x))


class MaskTwoStreamConvSwitchNet(nn.Module):
    """MaskTwoStreamConvSwitch_NET (--no_comb) and, with ``comb=True``, MaskTwoStreamConv_NET (the parser's default): the
    same modules, the latter returns the object-gated combination of the two streams' logits (:206-223) and has no
    dilated blocks."""

    def __init__(self, label_nc=35, output_nc=35, conv_dim=64, num_layers=3, conv_size=4, n_blocks=6,
                 cond_in='ctx_obj', which_stream='obj_context', align_corners=False, norm_layer='batch',
                 add_dilated_layers=False, comb=False, use_simpleRes=False):
        super().__init__()
        assert 'obj' in which_stream or 'context' in which_stream              # MaskTwoStreamConv_NET.py:18
        norm = _norm(norm_layer)
        self.num_layers = num_layers
        self.which_stream, self.comb = which_stream, comb
        input_nc = label_nc * 2 if cond_in == 'ctx_obj' else label_nc
        dims = [conv_dim, 96, 128, 256, 512]
        enc = [nn.Conv2d(input_nc, dims[0], 7, 2, 3), norm(dims[0]), nn.ReLU()]
        for i in range(num_layers):
            enc.append(DownResBlock3x3(dims[i], dims[i + 1], norm) if use_simpleRes else
                       ConvResnetBlock(dims[i], dims[i + 1], 2, conv_size, norm))
        self.conv_encoder_modules = nn.Sequential(*enc)
        latent = dims[num_layers]
        lat = []
        if add_dilated_layers and not comb:  # MaskTwoStreamConvSwitch_NET.py:103-105 (--add_dilated_layers, the ADE recipe)
            lat += [DilatedResnetBlock(latent, 2, norm), DilatedResnetBlock(latent, 4, norm)]
        lat += [ResnetBlock(latent, norm) for _ in range(int(math.floor(n_blocks / 2)))]
        self.latent_encoder = nn.Sequential(*lat)

        def decoder(out_nc, skip):
            layers, od = [], latent
            for i in range(num_layers + 1):
                idim = od
                od = dims[num_layers - i - 1] if i < num_layers else idim // 2
                if skip and 1 <= i <= num_layers:
                    idim *= 2
                layers.append(UpResBlock3x3(idim, od, norm) if use_simpleRes else
                              DeconvResnetBlock(idim, od, 2, conv_size, align_corners, norm))
            layers.append(nn.Conv2d(od, out_nc, 3, 1, 1))
            return nn.Sequential(*layers)

        def latent_dec():
            return nn.Sequential(*[ResnetBlock(latent, norm) for _ in range(int(math.ceil(n_blocks / 2)))])

        if 'obj' in which_stream:
            self.obj_conv_decoder_modules = decoder(1, False)
            self.obj_latent_decoder = latent_dec()
        if 'context' in which_stream:
            self.ctx_conv_decoder_modules = decoder(output_nc, True)
            self.ctx_latent_decoder = latent_dec()

    def _decode(self, dec, feat, skips):
        for i, layer in enumerate(dec):
            if skips is not None and 1 <= i <= self.num_layers:
                feat = torch.cat((skips[-i], feat), 1)
            feat = layer(feat)
        return feat

    def forward(self, x, cls_onehot=None):
        e = self.conv_encoder_modules
        f = e[2](e[1](e[0](x)))
        skips = []                          # what the decoder will find in enc_features after the in-place ReLUs
        for i in range(3, 3 + self.num_layers):
            f, r = e[i](f)
            skips.append(r)                 # = relu(previous stage output); for i = 3 it is the stem's ReLU output itself
        latent = self.latent_encoder(f)
        ctx_logit = ctx_prob = obj_logit = obj_prob = None
        if 'context' in self.which_stream:
            ctx_logit = self._decode(self.ctx_conv_decoder_modules, self.ctx_latent_decoder(latent), skips)
            ctx_prob = F.log_softmax(ctx_logit, 1)
        if 'obj' in self.which_stream:
            obj_logit = self._decode(self.obj_conv_decoder_modules, self.obj_latent_decoder(latent), None)
            obj_prob = torch.sigmoid(obj_logit)
        if not self.comb:
            return ctx_logit, ctx_prob, obj_logit, obj_prob
        comb_logit, comb_prob = ctx_logit, ctx_prob                          # MaskTwoStreamConv_NET.py:209-212
        if ctx_logit is not None and obj_logit is not None:                  # :213-221 (its bmm product is never used)
            gate = obj_prob.expand_as(ctx_prob)
            comb_logit = (1 - gate) * ctx_logit + gate * obj_logit
            comb_prob = F.log_softmax(comb_logit, 1)
        return comb_logit, comb_prob, obj_logit, obj_prob


# ------------------------------------------------------------------------------------------------------------------
# the box2mask training step: TwoStreamAE_mask.forward (reference models/TwoStreamAE_mask.py:167-254)
# ------------------------------------------------------------------------------------------------------------------
LOSS_NAMES = ['G_Recon_comb', 'G_Recon_obj', 'KL_loss', 'loss_G_GAN', 'loss_D_GAN', 'loss_G_GAN_Feat']


def encode_input(label_map, mask_ctx_in, mask_in, cls, label_nc):
    """TwoStreamAE_mask.encode_input (:127-152): one-hot of the ground-truth map, of the context map, and the object box
    mask written into the channel of the object's class."""
    B, _, H, W = label_map.shape
    gt = torch.zeros(B, label_nc, H, W).scatter_(1, label_map.long(), 1.0)
    ctx = torch.zeros(B, label_nc, H, W).scatter_(1, mask_ctx_in.long(), 1.0)
    obj = torch.zeros(B, label_nc, H, W)
    for b in range(B):
        obj[b, int(cls[b, 0])] = mask_in[b, 0]
    return gt, ctx, obj


def lr_control(loss_D_real, loss_D_fake, gan_margin=0.3):
    """Discriminator_NET.py:190-211: (g_lr, d_lr) in {0., 1.}: D pauses while either of its losses is under the margin,
    G pauses while either is above 1 - margin; if both would pause, both train."""
    update_d = not (loss_D_real < gan_margin or loss_D_fake < gan_margin)
    update_g = not (loss_D_real > 1 - gan_margin or loss_D_fake > 1 - gan_margin)
    if not (update_d or update_g):
        update_d = update_g = True
    return float(update_g), float(update_d)


class NLayerDiscriminator(nn.Module):
    """--which_gan patch: ONE PatchGAN (Discriminator_NET.py:60-125 with getIntermFeat=False: a flat ``model`` Sequential,
    keys ``model.<i>.*``) ending in a Sigmoid (use_sigmoid = not use_lsgan, TwoStreamAE_mask.py:69-76); forward(input, cond)
    concatenates the two."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='batch', use_sigmoid=True):
        super().__init__()
        norm = _norm(norm_layer)
        seq = [nn.Conv2d(input_nc, ndf, 4, 2, 2), nn.LeakyReLU(0.2)]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq += [nn.Conv2d(nf_prev, nf, 4, 2, 2), norm(nf), nn.LeakyReLU(0.2)]
        nf_prev, nf = nf, min(nf * 2, 512)
        seq += [nn.Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), nn.LeakyReLU(0.2), nn.Conv2d(nf, 1, 4, 1, 2)]
        if use_sigmoid:
            seq.append(nn.Sigmoid())
        self.model = nn.Sequential(*seq)

    def forward(self, x, cond):
        return self.model(torch.cat((x, cond), 1))


class NLayerResDiscriminator(nn.Module):
    """--which_gan patch_res (Discriminator_NET.py:118-183): the stride-2 stages are ConvResnetBlocks (kernel 4, LeakyReLU
    0.2 applied in place to the stage input, conv1x1 shortcut + norm), then the stride-1 block, the 1-channel head and the
    Sigmoid; flat ``model`` Sequential."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer='batch', use_sigmoid=True):
        super().__init__()
        norm = _norm(norm_layer)
        seq = [ConvResnetBlock(input_nc, ndf, 2, 4, norm, 0.2)]
        nf = ndf
        for _ in range(1, n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            seq.append(ConvResnetBlock(nf_prev, nf, 2, 4, norm, 0.2))
        nf_prev, nf = nf, min(nf * 2, 512)
        seq += [nn.Conv2d(nf_prev, nf, 4, 1, 2), norm(nf), nn.LeakyReLU(0.2), nn.Conv2d(nf, 1, 4, 1, 2)]
        if use_sigmoid:
            seq.append(nn.Sigmoid())
        self.model = nn.Sequential(*seq)

    def forward(self, x, cond):
        h = torch.cat((x, cond), 1)
        for layer in self.model:
            h = layer(h)
            h = h[0] if isinstance(h, tuple) else h          # ConvResnetBlock here returns (output, rectified input)
        return h


class TwoStreamAEMask(object):
    """CPU restatement of the reference trainer (models/TwoStreamAE_mask.py:14-254).  Defaults = the flags of
    scripts/train_box2mask_city.sh (which_stream obj_context, cond_in ctx_obj, use_gan patch_multiscale, objReconLoss bce,
    use_output_gate, use_ganFeat_loss, norm_layer batch, --no_comb); the parser's other values of those flags -- without
    --no_comb (MaskTwoStreamConv_NET), --which_stream obj | context, --cond_in ctx | obj, --which_gan patch | patch_res,
    --objReconLoss l1 | none, --use_simpleRes -- follow the same lines.  ``step`` = one
    TwoStreamAE_mask.forward(eval_mode=False): losses, then the G Adam step, then the D Adam step (:237-248)."""

    def __init__(self, label_nc=35, ndf=64, num_layers_D=3, gan_weight=0.1, rec_weight=1.0, lambda_feat=1.0, lr=0.0002,
                 beta1=0.5, beta2=0.999, use_output_gate=True, use_ganFeat_loss=True, norm_layer='batch',
                 add_dilated_layers=False, lr_control=False, no_comb=True, which_stream='obj_context', cond_in='ctx_obj',
                 which_gan='patch_multiscale', objReconLoss='bce', use_simpleRes=False):
        from oracle import ref_cpu
        self.label_nc, self.gan_weight, self.rec_weight, self.lambda_feat = label_nc, gan_weight, rec_weight, lambda_feat
        self.use_output_gate, self.use_ganFeat_loss, self.num_layers_D = use_output_gate, use_ganFeat_loss, num_layers_D
        self.lr_control, self.which_stream, self.cond_in = lr_control, which_stream, cond_in
        self.which_gan, self.objReconLoss = which_gan, objReconLoss
        self.netG = MaskTwoStreamConvSwitchNet(label_nc, label_nc, norm_layer=norm_layer, cond_in=cond_in,
                                               which_stream=which_stream, add_dilated_layers=add_dilated_layers,
                                               comb=not no_comb, use_simpleRes=use_simpleRes)
        d_nc = 1 + (2 * label_nc if cond_in == 'ctx_obj' else label_nc)      # :67-68
        if which_gan == 'patch':                                             # :69-76 (BCE on Sigmoid outputs)
            self.netD = NLayerDiscriminator(d_nc, ndf, num_layers_D, norm_layer, use_sigmoid=True)
        elif which_gan == 'patch_res':                                       # :77-84
            self.netD = NLayerResDiscriminator(d_nc, ndf, num_layers_D, norm_layer, use_sigmoid=True)
        elif which_gan == 'patch_multiscale':                                # :85-94 (LSGAN, 2 scales, features kept)
            self.netD = ref_cpu.MultiscaleDiscriminator(d_nc, ndf, num_layers_D, num_D=2, norm=norm_layer)
        else:
            raise NotImplementedError('which_gan [%s]: patch | patch_res | patch_multiscale' % which_gan)
        self.optimizer = torch.optim.Adam(self.netG.parameters(), lr=lr, betas=(beta1, beta2))
        self.optimizer_D = torch.optim.Adam(self.netD.parameters(), lr=lr, betas=(beta1, 0.999))

    def gan_loss(self, pred, real):
        """GANLoss.__call__ (losses.py:43-53).  patch_multiscale: a list of per-scale lists -> MSE summed over the scales.
        patch: ``pred`` is ONE tensor, so ``input[-1]`` there indexes the BATCH: the BCE of the LAST sample's patch map."""
        from oracle import ref_cpu
        if self.which_gan == 'patch_multiscale':
            return ref_cpu.gan_loss(pred, real)
        last = pred[-1]
        return F.binary_cross_entropy(last, torch.full_like(last, 1.0 if real else 0.0))

    def discriminate(self, x, cond):                                         # :154-159
        if self.which_gan == 'patch_multiscale':
            return self.netD(torch.cat((x, cond), 1))
        return self.netD(x, cond)

    def step(self, batch):
        label, mask_out, mask_in = batch['label'], batch['mask_out'], batch['mask_in']
        gt_onehot, ctx, obj = encode_input(label, batch['mask_ctx_in'], mask_in, batch['cls'], self.label_nc)
        cond = {'obj': obj, 'ctx': ctx, 'ctx_obj': torch.cat((obj, ctx), 1)}[self.cond_in]      # :341-347
        self.netG.train()
        _, comb_prob, _, obj_prob = self.netG(cond)
        obj_gt = batch['mask_obj_inst']
        loss_comb = loss_obj = torch.zeros(())
        if 'context' in self.which_stream:                                   # :190-191
            tgt = label[:, 0].long().clone()
            tgt[mask_out[:, 0] < 0.5] = 255                   # MaskReconLoss (mask_losses.py:20-26)
            loss_comb = F.nll_loss(comb_prob, tgt, ignore_index=255)
        if 'obj' not in self.which_stream:                                   # :280-281: a constant in the object's place
            obj_prob = torch.zeros_like(obj_gt)
        elif self.objReconLoss != 'none':                                    # :192-195
            if self.use_output_gate:
                obj_prob = obj_prob * mask_out
            loss_obj = (F.l1_loss if self.objReconLoss == 'l1' else F.binary_cross_entropy)(obj_prob, obj_gt)
        real, fake, dcond = obj_gt, obj_prob, cond
        if self.use_output_gate:
            real, fake, dcond = real * mask_out, fake * mask_out, dcond * mask_out
        real_d = self.discriminate(real, dcond)
        fake_d = self.discriminate(fake.detach(), dcond)
        loss_D_real, loss_D_fake = self.gan_loss(real_d, True), self.gan_loss(fake_d, False)
        loss_D = 0.5 * loss_D_real + 0.5 * loss_D_fake
        loss_feat = torch.zeros(1)
        if self.use_ganFeat_loss:                              # computed and returned, never added to loss_G (:225-227)
            fw, dw = 4.0 / (self.num_layers_D + 1), 1.0 / 2.0
            for i in range(2):                                 # patch: fake_d[i] is one sample's map, len 1 -> no term
                for j in range(len(fake_d[i]) - 1):
                    loss_feat = loss_feat + dw * fw * F.l1_loss(fake_d[i][j], real_d[i][j].detach()) * self.lambda_feat
        loss_G_GAN = self.gan_loss(self.discriminate(fake, dcond), True)
        loss_G = loss_obj + self.rec_weight * loss_comb + self.gan_weight * loss_G_GAN
        if self.lr_control:                                    # Discriminator_NET.py:190-211, TwoStreamAE_mask.py:229-245
            g_lr, d_lr = lr_control(float(loss_D_real.detach()), float(loss_D_fake.detach()))
            loss_G, loss_D = g_lr * loss_G, d_lr * loss_D       # a frozen net still takes an Adam step on zero gradients
        self.optimizer.zero_grad()
        loss_G.backward()
        self.optimizer.step()
        self.optimizer_D.zero_grad()
        loss_D.backward()
        self.optimizer_D.step()
        vals = [loss_comb, loss_obj, 0.0, loss_G_GAN, loss_D, loss_feat]
        return dict(zip(LOSS_NAMES, [float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else v for v in vals]))
