"""box2mask trainer on the HIP kernels: the reference's ``TwoStreamAE_mask`` (models/TwoStreamAE_mask.py).  Defaults = the
configuration of scripts/train_box2mask_city.sh (``--model AE_maskgen_twostream --no_comb --which_stream obj_context
--cond_in ctx_obj --use_gan --which_gan patch_multiscale --objReconLoss bce --norm_layer batch --use_output_gate
--use_ganFeat_loss``); the parser's other values of those flags run too (round 6: without --no_comb, --which_stream obj |
context, --cond_in obj | ctx, --which_gan patch | patch_res, --objReconLoss l1 | none, --use_simpleRes).
``forward(..., eval_mode=False)`` IS the training step, as in the reference (:167-254): losses, then the generator's Adam
step, then the discriminator's, all inside the call.

Departures that do not change the arithmetic: the three one-hot tensors are built straight into the 70-channel condition
buffer; ``x * mask.repeat`` + ``torch.cat`` are one kernel; the discriminator pass on the attached fake runs with frozen D
weights (the reference computes those weight gradients and zeroes them before loss_D.backward(), :237-247); both Adams
are fused flat-arena steps."""
import argparse
from collections import OrderedDict

import torch

from .. import ops
from ..nn import frozen_params
from ..optim import FusedAdam
from .base_model import BaseModel
from .Discriminator_NET import MultiscaleDiscriminator, NLayerDiscriminator, NLayerResDiscriminator
from .layer_util import torch_default_init
from .MaskTwoStreamConvSwitch_NET import MaskTwoStreamConvSwitch_NET, MaskTwoStreamConv_NET
from .pix2pixHD_condImg_model import pick_device

DEFAULTS = dict(label_nc=35, output_nc=35, num_layers=3, conv_dim=64, conv_size=4, norm_layer='batch', n_blocks=6,
                which_stream='obj_context', cond_in='ctx_obj', use_gan=True, which_gan='patch_multiscale', gan_weight=0.1,
                rec_weight=1.0, use_output_gate=True, ndf=64, num_layers_D=3, objReconLoss='bce', use_ganFeat_loss=True,
                lambda_feat=1.0, lr=0.0002, beta1=0.5, beta2=0.999, lr_control=False, no_comb=True, isTrain=True,
                gpu_ids=[0], checkpoints_dir='./checkpoints', name='box2mask', niter=400, niter_decay=0)


def complete(opt):
    if isinstance(opt, dict):
        opt = argparse.Namespace(**opt)
    for k, v in DEFAULTS.items():
        if not hasattr(opt, k):
            setattr(opt, k, v)
    return opt


class TwoStreamAE_mask(BaseModel):
    def name(self):
        return 'TwoStreamAE_mask'

    def __init__(self, opt):
        opt = complete(opt)
        super().__init__(opt)
        # every value the parser offers for these flags runs (round 6); what the reference itself cannot run fails here
        if opt.cond_in not in ('obj', 'ctx', 'ctx_obj'):
            raise NotImplementedError('--cond_in [%s]: obj | ctx | ctx_obj (reference construct_input_cond)' % opt.cond_in)
        if opt.use_gan and opt.which_gan not in ('patch', 'patch_res', 'patch_multiscale'):
            # the reference builds no discriminator for any other value and fails at its first use (:69-94)
            raise NotImplementedError('--which_gan [%s]: patch | patch_res | patch_multiscale' % opt.which_gan)
        if opt.isTrain and not opt.use_gan:
            # the reference's forward dies on its first step without --use_gan (loss_G_GAN_Feat is only bound inside
            # ``if self.use_gan``, TwoStreamAE_mask.py:251): refuse at construction instead of training something else
            raise NotImplementedError('box2mask without --use_gan: the reference fails in its first step (TwoStreamAE_mask.py:251)')
        self.device = pick_device(opt)
        self.use_gan, self.use_output_gate = bool(opt.use_gan), bool(opt.use_output_gate)
        self.which_stream, self.cond_in, self.which_gan = opt.which_stream, opt.cond_in, opt.which_gan
        # --objReconLoss l1 | bce, anything else: no object reconstruction term (reference :50-55)
        self.objReconLoss = opt.objReconLoss if opt.objReconLoss in ('l1', 'bce') else None
        net = MaskTwoStreamConvSwitch_NET if opt.no_comb else MaskTwoStreamConv_NET      # reference :29-32
        self.netG = torch_default_init(net(opt)).to(self.device)   # never weights_init'ed upstream
        self.loss_names = ['G_Recon_comb', 'G_Recon_obj', 'KL_loss', 'loss_G_GAN', 'loss_D_GAN', 'loss_G_GAN_Feat']
        self.reducer_G = self.reducer_D = None     # set by dist.attach_data_parallel (one process per GPU)
        if self.isTrain:
            self.old_lr = opt.lr
            self.optimizer = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, opt.beta2))
            if self.use_gan:
                d_nc = 1 + (2 * opt.label_nc if opt.cond_in == 'ctx_obj' else opt.label_nc)       # reference :67-68
                if opt.which_gan == 'patch':       # one PatchGAN with a Sigmoid, BCE (reference :69-76)
                    self.netD = NLayerDiscriminator(d_nc, opt.ndf, opt.num_layers_D, opt.norm_layer, True, False)
                elif opt.which_gan == 'patch_res':  # the same with ConvResnetBlock stages (:77-84)
                    self.netD = NLayerResDiscriminator(d_nc, opt.ndf, opt.num_layers_D, opt.norm_layer, True, False)
                else:                              # LSGAN on two scales, intermediate features kept (:85-94)
                    self.netD = MultiscaleDiscriminator(d_nc, opt.ndf, opt.num_layers_D, opt.norm_layer, False, 2, True)
                self.netD.to(self.device)
                self.optimizer_D = FusedAdam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))

    @property
    def optimizer_G(self):
        """alias used by dist.attach_data_parallel (the reference calls the generator's optimizer ``optimizer``)."""
        return self.optimizer

    def _dev(self, t):
        return None if t is None else t.to(self.device, dtype=torch.float32).contiguous()

    # -- TwoStreamAE_mask.encode_input (:127-152) + construct_input_cond (:353-360) ---------------------------------
    def encode_cond(self, mask_ctx_in, mask_in, cls):
        """The generator / discriminator condition of --cond_in (reference :341-347): 'ctx_obj' cat(one-hot box mask in the
        object's class channel, one-hot(context label map)) -> (B, 2*label_nc, H, W); 'obj' / 'ctx' one half of it."""
        nc = self.opt.label_nc
        ctx, box = self._dev(mask_ctx_in), self._dev(mask_in)
        B, _, H, W = ctx.shape
        halves = {'ctx_obj': 2, 'obj': 1, 'ctx': 1}[self.cond_in]
        cond = torch.empty((B, halves * nc, H, W), dtype=torch.float32, device=self.device)
        from .._cabi import lib
        st = torch.cuda.current_stream().cuda_stream
        # class ids travel to the device once (B floats, asynchronous); both halves are written by one kernel each --
        # no host read-back of cls, no per-sample copies, no zero fill
        if self.cond_in != 'ctx':
            cls_dev = cls.reshape(-1).to(self.device, torch.float32, non_blocking=True).contiguous()
            lib.him_class_mask(box.data_ptr(), cls_dev.data_ptr(), cond.data_ptr(), B, nc, halves * nc, 0, H * W, st)
        if self.cond_in != 'obj':
            lib.him_onehot(ctx.data_ptr(), cond.data_ptr(), B, nc, halves * nc, (halves - 1) * nc, H * W, st)
        return cond

    def _gan(self, preds, real):
        """GANLoss.__call__ (reference losses.py:43-53).  patch_multiscale: MSE of each scale's last output, summed.
        patch: ``preds`` is ONE (B,1,h,w) tensor and ``input[-1]`` there indexes the BATCH -- the BCE of the LAST sample's
        patch map against the constant target, reproduced as it is."""
        if self.which_gan != 'patch_multiscale':
            last = preds[-1:].contiguous()
            return ops.bce_mean(last, torch.full_like(last, 1.0 if real else 0.0))
        loss = 0
        for p in preds:
            loss = loss + ops.mse_const(p[-1], 1.0 if real else 0.0)
        return loss

    def forward(self, label_map, mask_obj_in, mask_ctx_in, mask_obj_out, mask_out, mask_obj_inst, cls, mask_in,
                eval_mode=False):
        opt = self.opt
        cond = self.encode_cond(mask_ctx_in, mask_in, cls)
        gate = self._dev(mask_out)
        self.netG.train(not eval_mode)
        _, comb_prob, _, obj_prob = self.netG(cond)
        if eval_mode:
            return self._labels(comb_prob, obj_prob, label_map, gate)
        label = self._dev(label_map)
        zero = torch.zeros((), device=self.device)
        obj_gt = self._dev(mask_obj_inst)
        loss_comb = ops.masked_nll(comb_prob, label, gate) if 'context' in self.which_stream else zero      # :190-191
        loss_obj = zero
        if 'obj' not in self.which_stream:
            obj_prob = torch.zeros_like(obj_gt)       # a constant stands in for the object stream (reference :280-281)
        obj_recon_label = obj_prob.detach()          # reconstruct()'s 'obj_recon_label': the UNGATED probability (:280-283)
        if 'obj' in self.which_stream and self.objReconLoss is not None:                                    # :192-195
            if self.use_output_gate:
                obj_prob = ops.mul_mask(obj_prob, gate)
            loss_obj = (ops.l1_mean if self.objReconLoss == 'l1' else ops.bce_mean)(obj_prob, obj_gt)
        loss_G_GAN, loss_D, loss_feat = zero, zero, torch.zeros(1, device=self.device)
        if self.use_gan:
            m = gate if self.use_output_gate else None
            real_d = self.netD(ops.cat_channels([obj_gt, cond], m, 1))
            fake_d = self.netD(ops.cat_channels([obj_prob.detach(), cond], m, 1))
            self._d_real_loss, self._d_fake_loss = self._gan(real_d, True), self._gan(fake_d, False)
            loss_D = 0.5 * self._d_real_loss + 0.5 * self._d_fake_loss
            if opt.use_ganFeat_loss and self.which_gan == 'patch_multiscale':
                # returned, never added to loss_G (reference :225-227); with --which_gan patch the reference's double loop
                # walks one sample's (1,h,w) map and finds no feature pair: the term stays 0
                with torch.no_grad():
                    fw, dw = 4.0 / (opt.num_layers_D + 1), 1.0 / 2.0
                    for i in range(2):
                        for j in range(len(fake_d[i]) - 1):
                            loss_feat = loss_feat + dw * fw * ops.l1_mean(fake_d[i][j], real_d[i][j]) * opt.lambda_feat
            with frozen_params():
                loss_G_GAN = self._gan(self.netD(ops.cat_channels([obj_prob, cond], m, 1)), True)
        loss_G = loss_obj + opt.rec_weight * loss_comb + opt.gan_weight * loss_G_GAN
        if self.use_gan and opt.lr_control:
            # reference :229-245 with the predicate of Discriminator_NET.py:190-211 evaluated ON THE DEVICE (the
            # reference reads three losses back to the host every step): loss_G / loss_D are scaled by g_lr / d_lr in
            # {0, 1}; a "frozen" net still takes its Adam step on zero gradients, exactly as upstream
            g_lr, d_lr = ops.lr_control(self._d_real_loss, self._d_fake_loss)
            loss_G, loss_D = loss_G * g_lr, loss_D * d_lr
        self.optimizer.zero_grad()
        if self.reducer_G is not None:
            self.reducer_G.begin()
        loss_G.backward()
        if self.reducer_G is not None:
            self.reducer_G.finish()
        self.optimizer.step()
        if self.use_gan:
            self.optimizer_D.zero_grad()
            if self.reducer_D is not None:
                self.reducer_D.begin(contributions=2)   # D(real) and D(fake.detach()) both reach D's weights
            loss_D.backward()
            if self.reducer_D is not None:
                self.reducer_D.finish()
            self.optimizer_D.step()
        # [comb_recon_label, obj_recon_label] as train_box2mask.py:70-71 reads them (the label map is an int64 arg-max map)
        comb_label = (self._comb_label(comb_prob.detach(), label, gate) if comb_prob is not None
                      else torch.zeros_like(gate))          # reference :278-279 without the context stream
        return [loss_comb.detach(), loss_obj.detach(), 0, loss_G_GAN.detach(), loss_D.detach(), loss_feat.detach()], \
               [comb_label, obj_recon_label]

    @staticmethod
    def _comb_label(comb_prob, label, gate):
        """arg-max over channels of postprocess_output(log-prob, mask, one-hot(gt)) for a BINARY mask, as an int64
        (B,1,H,W) map like torch.max's indices (:276-277): the arg-max log-probability inside the box, the ground-truth
        label outside (there the blended map IS the one-hot) -- without materialising the (B,label_nc,H,W) blend."""
        with torch.no_grad():
            return torch.where(gate >= 0.5, comb_prob.argmax(1, keepdim=True), label.long())

    def _labels(self, comb_prob, obj_prob, label_map, gate):
        """generate()'s outputs (:298-301): reconstruct() in eval mode."""
        comb, obj = self._stream_outputs(comb_prob, obj_prob, gate)
        return {'comb_pred_label': comb if comb_prob is None else self._comb_label(comb_prob, self._dev(label_map), gate),
                'obj_pred_label': obj.detach()}

    @staticmethod
    def _stream_outputs(comb_prob, obj_prob, gate):
        """What reconstruct() puts in a missing stream's place (reference :278-281): zeros shaped like the mask."""
        return (torch.zeros_like(gate) if comb_prob is None else comb_prob,
                torch.zeros_like(gate) if obj_prob is None else obj_prob)

    def generate(self, input_dict):
        """Reference :298-301: reconstruct() in eval mode (running BatchNorm statistics; the generator's training /
        eval mode is what it was afterwards)."""
        out = self.reconstruct(input_dict, eval_mode=True)
        return {'comb_pred_label': out['comb_recon_label'], 'obj_pred_label': out['obj_recon_label']}

    # -- the reference's remaining public methods (evaluation / visualisation scripts and user code call them) ----------
    def get_model(self, model_factory):
        print(self.name())
        return model_factory

    def encode_input(self, label_map, mask_ctx_in, mask_out, mask_in, cls, infer=False):
        """-> (one-hot label map, one-hot context map, mask_out, one-hot class (B,label_nc), the box mask in the object's
        class channel), reference :127-152, each written by one kernel (him_onehot / him_class_mask write every channel:
        no zero fill; the reference leaves its class one-hot uninitialised outside the hot entry, :139-140)."""
        from .._cabi import lib
        nc = self.opt.label_nc
        label, ctx = self._dev(label_map), self._dev(mask_ctx_in)
        B, _, H, W = label.shape
        st = torch.cuda.current_stream().cuda_stream
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)  # noqa: E731
        onehot_label, onehot_ctx, cls_onehot = new(B, nc, H, W), new(B, nc, H, W), new(B, nc)
        cls_dev = cls.reshape(-1).to(self.device, torch.float32).contiguous()
        lib.him_onehot(label.data_ptr(), onehot_label.data_ptr(), B, nc, nc, 0, H * W, st)
        lib.him_onehot(ctx.data_ptr(), onehot_ctx.data_ptr(), B, nc, nc, 0, H * W, st)
        lib.him_onehot(cls_dev.data_ptr(), cls_onehot.data_ptr(), B, nc, nc, 0, 1, st)
        if mask_in is None:
            obj_cond = torch.zeros((B, nc, H, W), dtype=torch.float32, device=self.device)
        else:
            obj_cond = new(B, nc, H, W)
            lib.him_class_mask(self._dev(mask_in).data_ptr(), cls_dev.data_ptr(), obj_cond.data_ptr(), B, nc, nc, 0, H * W, st)
        return onehot_label, onehot_ctx, self._dev(mask_out), cls_onehot, obj_cond

    def construct_input_cond(self, obj_cond, ctx_cond):
        """Reference :353-360."""
        if self.opt.cond_in == 'obj':
            return obj_cond
        if self.opt.cond_in == 'ctx':
            return ctx_cond
        return ops.cat_channels([obj_cond, ctx_cond])

    def discriminate(self, input, cond):
        """Reference :154-159 ('patch_multiscale': the condition is concatenated behind the mask)."""
        return self.netD(ops.cat_channels([input, cond]))

    def mask_variable(self, input, mask):
        """input * mask.repeat(1, C, 1, 1) (reference :161-165)."""
        if input.dim() != mask.dim():
            mask = mask.unsqueeze(1)
        return ops.mul_mask(input, mask.contiguous())

    def postprocess_output(self, prob_map, gt_mask, gt_one_hot, use_blending=False):
        """prob_map * gt_mask + (1 - gt_mask) * gt_one_hot (reference :363-371), one kernel."""
        return ops.blend(gt_one_hot, prob_map, self._dev(gt_mask))

    def reconstruct(self, input_dict, eval_mode=False):
        """Reference :257-296: one generator pass WITHOUT a training step; ``eval_mode`` runs it on the running BatchNorm
        statistics (and without a tape), otherwise in training mode with the tape -- the generator's mode is restored."""
        label_map, cls = input_dict['label_map'], input_dict['cls']
        was_training = self.netG.training
        self.netG.train(not eval_mode)
        try:
            gt_one_hot, input_ctx, gt_mask, _, input_obj_cond = self.encode_input(
                label_map, input_dict['mask_ctx_in'], input_dict['mask_out'], input_dict['mask_in'], cls)
            cond = self.construct_input_cond(input_obj_cond, input_ctx)
            with torch.set_grad_enabled(not eval_mode and torch.is_grad_enabled()):
                _, comb_prob, _, obj_prob = self.netG(cond)
                if comb_prob is not None:
                    comb_onehot = self.postprocess_output(comb_prob, gt_mask, gt_one_hot)
            comb_label = (comb_onehot.detach().argmax(1, keepdim=True) if comb_prob is not None
                          else torch.zeros_like(gt_mask))
            if obj_prob is None:
                obj_prob = torch.zeros_like(gt_mask)
        finally:
            self.netG.train(was_training)
        out = {'comb_recon_label': comb_label, 'obj_recon_label': obj_prob}
        if not eval_mode:
            out.update({'label_map': self._dev(label_map), 'input_mask': input_ctx, 'comb_gt_mask': gt_mask,
                        'obj_gt_mask': self._dev(input_dict['mask_obj_inst']), 'comb_recon_prob': comb_prob,
                        'obj_recon_prob': obj_prob, 'input_obj_cond': input_obj_cond})
        return out

    def evaluate(self, input_dict, target_size=None):
        """Reference :303-348 for the first sample of the batch: the label map with the predicted object pasted in (or, for
        the background class ``label_nc - 1``, the arg-max context map inside the box).  ``target_size`` (bilinear resize
        of the probabilities to the original resolution) is the visualisation scripts' path and not built."""
        if target_size is not None:
            raise NotImplementedError('TwoStreamAE_mask.evaluate(target_size=...): arbitrary-size bilinear resampling is '
                                      'not on the HIP path')
        first = lambda k: input_dict[k][0].unsqueeze(0)  # noqa: E731
        label_map, cls = first('label_map'), first('cls')
        was_training = self.netG.training
        self.netG.train(False)
        try:
            with torch.no_grad():
                gt_one_hot, input_ctx, gt_mask, _, input_obj_cond = self.encode_input(
                    label_map, first('mask_ctx_in'), first('mask_out'), first('mask_in'), cls)
                _, comb_prob, _, obj_prob = self.netG(self.construct_input_cond(input_obj_cond, input_ctx))
                if comb_prob is None or obj_prob is None:
                    raise NotImplementedError('evaluate() needs both streams (the reference reads both, :316-334)')
                if self.use_output_gate:
                    obj_prob = self.mask_variable(obj_prob, gt_mask)
                cls_id = int(cls.reshape(-1)[0])
                if cls_id == self.opt.label_nc - 1:
                    return self.postprocess_output(comb_prob, gt_mask, gt_one_hot).argmax(1, keepdim=True)
                obj_mask = (obj_prob > 0.5).float()
                return (1 - obj_mask) * self._dev(label_map) + obj_mask * float(cls_id)
        finally:
            self.netG.train(was_training)

    def delete_model(self, which_epoch):
        """Reference :366-369 (train_box2mask.py:142 rotates checkpoints with it)."""
        self.delete_network('G', which_epoch, self.gpu_ids)
        if self.use_gan:
            self.delete_network('D', which_epoch, self.gpu_ids)

    # -- checkpoints: the reference's per-module dict (base_model.py:52-66) ------------------------------------------
    @property
    def params_dict(self):
        g, d = self.netG, OrderedDict()
        for i, m in enumerate(g.conv_encoder_modules):
            d['conv_encoder_%d' % i] = m
        d['latent_encoder'] = g.latent_encoder
        if 'obj' in g.which_stream:
            for i, m in enumerate(g.obj_conv_decoder_modules):
                d['obj_conv_decoder_%d' % i] = m
            d['obj_latent_decoder'] = g.obj_latent_decoder
        if 'context' in g.which_stream:
            for i, m in enumerate(g.ctx_conv_decoder_modules):
                d['ctx_conv_decoder_%d' % i] = m
            d['ctx_latent_decoder'] = g.ctx_latent_decoder
        return d

    def save(self, which_epoch):
        self.save_network_dict(self.params_dict, self.optimizer, 'G', which_epoch, self.gpu_ids)
        if self.use_gan:
            self.save_network(self.netD, 'D', which_epoch, self.gpu_ids)

    def update_learning_rate(self, epoch=0, data_size=0):
        if epoch > self.opt.niter:
            lr = self.old_lr - self.opt.lr / self.opt.niter_decay
            for g in self.optimizer.param_groups + (self.optimizer_D.param_groups if self.use_gan else []):
                g['lr'] = lr
            self.old_lr = lr
