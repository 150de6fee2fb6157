"""``Pix2PixHDModel_condImg`` on the MI355X kernels: the reference trainer object
(``models/pix2pixHD_condImg_model.py:23-327``) with the same constructor flags, attributes
(``loss_names``, ``optimizer_G/D``, ``netG/netD``), ``forward`` signature and return value.

Added by this build (named by the task, absent from the reference): ``backward_G`` / ``backward_D`` /
``optimize_parameters`` -- thin wrappers around exactly the lines ``train_mask2image.py:68-86`` runs inline.
"""
import contextlib
import os
import random
from collections import OrderedDict

import numpy as np
import torch

from .. import ops, synth
from ..config import SCHED
from ..nn import frozen_params
from ..optim import FusedAdam
from .base_model import BaseModel

NULLVAL = 0.0
# Schedule switches live in config.SCHED (run-time object; defaults = the shipped schedule).  vgg_backward_early (off):
# VGG's backward started on its stream BEFORE loss_D.backward() instead of behind it (see optimize_parameters) -- removes a
# 5.4 ms window in which only the VGG stream works, and costs 0.7 ms per step (60.2 vs 59.5 ms, three repetitions): the
# fused-Winograd kernels use the chip well alone and slow D's backward when they share it.


class ImagePool(object):
    """History buffer of generated images (reference ``util/image_pool.py``); identity when pool_size == 0,
    which is what every shipped recipe uses."""

    def __init__(self, pool_size):
        self.pool_size, self.images = pool_size, []

    def query(self, images):
        if self.pool_size == 0:
            return images
        out = []
        for image in images.detach():
            image = image.unsqueeze(0)
            if len(self.images) < self.pool_size:
                self.images.append(image)
                out.append(image)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.pool_size - 1)
                out.append(self.images[i].clone())
                self.images[i] = image
            else:
                out.append(image)
        return torch.cat(out, 0)


def pick_device(opt):
    if not torch.cuda.is_available():
        raise RuntimeError('the mask2image HIP path needs an MI355X (no CUDA/HIP device visible); '
                           'there is no CPU fallback')
    if 'LOCAL_RANK' in os.environ:
        return torch.device('cuda', int(os.environ['LOCAL_RANK']))
    return torch.device('cuda', opt.gpu_ids[0] if len(opt.gpu_ids) else 0)


class Pix2PixHDModel_condImg(BaseModel):
    color = False

    def __init__(self, opt):
        super().__init__(opt)
        self.isTrain = opt.isTrain
        self.netG_type = opt.netG
        if opt.instance_feat or opt.label_feat:
            raise NotImplementedError('instance/label feature encoder (netE) is broken in the reference '
                                      '(Pix2Pix_NET.py:255) and outside the hot path')
        self.device = pick_device(opt)
        input_nc = opt.label_nc if opt.label_nc != 0 else 3
        netG_input_nc = input_nc + (0 if opt.no_instance else 1)
        self.n_label = netG_input_nc

        from .Pix2Pix_NET import GlobalGenerator, GlobalTwoStreamGenerator, LocalEnhancer
        if opt.netG == 'global':
            self.netG = GlobalGenerator(netG_input_nc + 3, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                        opt.n_blocks_global, opt.norm, 'reflect', opt.use_output_gate)
        elif opt.netG == 'global_twostream':
            self.netG = GlobalTwoStreamGenerator(netG_input_nc, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                                 opt.n_blocks_global, opt.norm, 'reflect', opt.use_skip,
                                                 opt.which_encoder, opt.use_output_gate, opt.feat_fusion,
                                                 extra_embed=self.color)
        elif opt.netG == 'local':      # reachable here, dead in the reference
            self.netG = LocalEnhancer(netG_input_nc + 3, opt.output_nc, opt.ngf, opt.n_downsample_global,
                                      opt.n_blocks_global, opt.n_local_enhancers, opt.n_blocks_local, opt.norm)
        else:
            raise NameError('global generator name is not defined properly: %s' % opt.netG)
        self.netG.to(self.device)
        if opt.netG == 'global':
            # the down-convolutions' weight gradients are the last of the backward pass (config.SCHED.g_tail_wgrad_alt)
            from ..nn import Conv2d as _HimConv2d, ConvTranspose2d as _HimDeconv2d
            for m in self.netG.model:
                if isinstance(m, _HimConv2d) and getattr(m, 'stride', 1) in (2, (2, 2)):
                    m.weight._him_wgrad_alt = 'tail'
                elif isinstance(m, _HimDeconv2d):
                    m.weight._him_wgrad_alt = 'head'     # config.SCHED.g_head_wgrad_alt

        if self.isTrain:
            self.no_imgCond = opt.no_imgCond
            self.mask_gan_input = opt.mask_gan_input
            self.use_soft_mask = opt.use_soft_mask
            netD_input_nc = input_nc + opt.output_nc + (0 if self.no_imgCond else 3)
            if not opt.no_instance:
                netD_input_nc += 1
            if opt.netG == 'global_twostream' and opt.which_encoder == 'ctx':
                netD_input_nc = 3
            from .Discriminator_NET import MultiscaleDiscriminator
            self.netD = MultiscaleDiscriminator(netD_input_nc, opt.ndf, opt.n_layers_D, opt.norm, opt.no_lsgan,
                                                opt.num_D, not opt.no_ganFeat_loss, spectral_norm=bool(getattr(opt, 'sn_D', False)))
            self.netD.to(self.device)

        if not self.isTrain or opt.continue_train or opt.load_pretrain:
            pretrained_path = '' if not self.isTrain else opt.load_pretrain
            self.load_network(self.netG, 'G', opt.which_epoch, pretrained_path)
            if self.isTrain:
                self.load_network(self.netD, 'D', opt.which_epoch, pretrained_path)

        if self.isTrain:
            if opt.pool_size > 0 and int(os.environ.get('WORLD_SIZE', '1')) > 1:
                raise NotImplementedError('Fake Pool Not Implemented for MultiGPU')
            self.fake_pool = ImagePool(opt.pool_size)
            self.old_lr = opt.lr
            from .losses import GANLoss, VGGLoss
            self.criterionGAN = GANLoss(use_lsgan=not opt.no_lsgan)
            self.criterionFeat = ops.l1_mean
            if not opt.no_vgg_loss:
                self.criterionVGG = VGGLoss(self.gpu_ids)
                if opt.vgg_weights:
                    self.criterionVGG.vgg.load_torchvision_state_dict(torch.load(opt.vgg_weights, map_location='cpu'))
                else:   # no network, no torchvision: seeded synthetic weights (He-normal), see DESIGN.md
                    self.criterionVGG.vgg.load_state_dict(
                        synth.init_state_dict(self.criterionVGG.vgg.state_dict(), 3, 'vgg'))
                self.criterionVGG.to(self.device)
            self.loss_names = ['G_GAN', 'G_GAN_Feat', 'G_VGG', 'D_real', 'D_fake']
            if opt.niter_fix_global > 0:
                # only the local enhancer trains at first (reference :122-130): one group per parameter, lr 0 elsewhere
                if getattr(opt, 'verbose', False):
                    print('------------- Only training the local enhancer network (for %d epochs) ------------'
                          % opt.niter_fix_global)
                params = [{'params': [value], 'lr': opt.lr if key.startswith('model' + str(opt.n_local_enhancers)) else 0.0}
                          for key, value in self.netG.named_parameters()]
            else:
                params = list(self.netG.parameters())
            self.optimizer_G = FusedAdam(params, lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D = FusedAdam(self.netD.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
            self.reducer_G = self.reducer_D = None
            self.comm_timing = None
            self._g_chunked = False
            if SCHED.adam_chunked:
                # one rank: a LOCAL reducer (no exchange) that only tracks which 64 MB gradient buckets are final and
                # triggers their Adam step (_bucket_update_G); dist.attach_data_parallel replaces it by the real one
                from ..dist import GradReducer
                arena = self.optimizer_G.arena
                self.reducer_G = GradReducer(arena.grad, [p._him_arena_range for p in arena.params], local=True,
                                             comm_stream=ops._opt_stream(self.device))
                self.reducer_G.attach(arena.params)
                self.reducer_G.bucket_hook = self._bucket_update_G
            # autograd creates a parameter's AccumulateGrad node lazily, on the stream that is current at the parameter's
            # first use of a step -- for D the side stream (real-image branch) -- and makes the caller's stream wait for
            # every such "leaf stream" when backward() returns: loss_G.backward() ended with the main stream waiting for
            # ALL weight gradients queued on the side stream (3.4 ms idle per step in the r02 trace) although nothing on
            # the main stream needs them (the optimizers join that stream themselves).  Holding the nodes, created here
            # on the main stream, pins them to it.
            with torch.cuda.device(self.device):
                self._grad_accumulators = [p.expand_as(p).grad_fn.next_functions[0][0]
                                           for p in list(self.netG.parameters()) + list(self.netD.parameters())
                                           if p.requires_grad]
            # gradient routing of the shared fake-image discriminator pass (see forward)
            self._d_weight_ids = set(id(p) for p in self.netD.parameters())
            self._d_first_weight_ids = set(id(getattr(self.netD, 'scale%d_layer0' % i)[0].weight)
                                           for i in range(opt.num_D))
            self._fake_gate = None
            self._share_fake_pass = False
        self.loss_G = self.loss_D = None

    def name(self):
        return 'Pix2PixHDModel_condImg'

    # ------------------------------------------------------------------------------------------
    def _dev(self, t):
        if t is None:
            return None
        t = t.detach()
        if t.dtype == torch.uint8:
            # compact id maps (label / instance): 1 byte per pixel over PCIe, widened to float ids on the device
            return ops.widen_u8(t.to(self.device, non_blocking=True))
        if t.device != self.device or t.dtype != torch.float32:
            t = t.to(self.device, torch.float32, non_blocking=True)
        return t.contiguous()

    def _color_embedding(self, obj_mask, real_image, color_embed, infer):
        return None

    def encode_input(self, label_map, inst_map=None, real_image=None, feat_map=None, mask_in=None, infer=False,
                     obj_mask=None, color_embed=None, lazy=False):
        """-> (input_label, inst_map, real_image, feat_map, cond_image), as the reference (:144-174); the
        concatenated [label | cond] buffer the generator/discriminator read is kept in ``self._enc``.
        ``lazy`` (the trainer's own calls): the one-hot block stays an id map -- ``input_label`` and the buffer are
        ``ops.LabelCond`` objects, which the stems, the first PatchGAN convolution and the pooled scales read directly; a
        caller of the public method gets the reference's tensors."""
        assert real_image is not None
        assert mask_in is not None
        label_map, inst_map = self._dev(label_map), self._dev(inst_map)
        real_image, mask_in = self._dev(real_image), self._dev(mask_in)
        emb = self._color_embedding(self._dev(obj_mask), real_image, color_embed, infer)
        buf, n_label, n_cond = ops.encode_channels(label_map, inst_map, real_image, mask_in, self.opt.label_nc,
                                                   not self.opt.no_instance, color_emb=emb,
                                                   lazy=lazy and SCHED.label_ids and self.opt.label_nc > 0)
        self._enc = (buf, n_label, n_cond, mask_in)
        input_label = ops.slice_channels(buf, 0, n_label)
        cond_image = ops.slice_channels(buf, n_label, n_cond)
        return input_label, inst_map, real_image, feat_map, cond_image

    def forward_wrapper(self, data, infer=False):
        """Batch dict -> ``forward`` (reference :188-196; train_mask2image.py:57 keeps the call commented out)."""
        return self.forward(data['label'], data['inst'], data['image'], None, data['mask_in'], data['mask_out'], infer)

    def _d_stateful(self):
        """True when a discriminator forward changes persisted state: --sn_D (power-iteration vectors) or --norm batch
        in training (BatchNorm running statistics, three updates per step in the reference: fake detached, real, fake)."""
        return bool(getattr(self.opt, 'sn_D', False)) or getattr(self.opt, 'norm', 'instance') == 'batch'

    def _d_split(self):
        """True when the discriminators get (condition, image) as an ``ops.CondImage`` pair instead of their concatenation:
        no mask on the input, no image pool (it stores concatenated tensors), a condition at all."""
        ctx_only = self.opt.netG == 'global_twostream' and self.opt.which_encoder == 'ctx'
        return SCHED.d_split_input and not ctx_only and not self.mask_gan_input and self.opt.pool_size == 0

    def _d_input(self, cond, image, mask):
        if self.opt.netG == 'global_twostream' and self.opt.which_encoder == 'ctx':
            return ops.mul_mask(image, mask) if self.mask_gan_input else image
        if self._d_split():
            return ops.CondImage(cond, image)       # condition and image kept apart (see ops.CondImage)
        return ops.cat_channels([cond, image], mask if self.mask_gan_input else None, 1)

    def discriminate(self, input_label, test_image, mask, use_pool=False):
        x = self._d_input(input_label, test_image.detach(), mask)
        if use_pool:
            x = self.fake_pool.query(x)
        return self.netD.forward(x)

    def _real_branch_ahead(self, netD_cond, real_image, mask_cond, inputs_ready=None):
        """D(real) + LSGAN loss and VGG(real) on the side stream (None when disabled).  ``inputs_ready``: event recorded
        on the main stream right after the input encoding -- the side stream waits for THAT, not for whatever the caller
        has enqueued on the main stream since (the generator forward)."""
        if not SCHED.real_ahead:
            return None
        main = torch.cuda.current_stream(self.device)
        side = ops._real_stream(self.device)
        if inputs_ready is not None:
            side.wait_event(inputs_ready)
        else:
            side.wait_stream(main)
        out = {'stream': side, 'y_vgg': None}
        # --sn_D / --norm batch: every discriminator forward moves persisted state (power-iteration vectors / BatchNorm
        # running statistics), so the three passes must run in the reference's order (fake-detached, real, fake): only
        # the VGG features of the real image run ahead
        with_d = not self._d_stateful()
        vgg_first = SCHED.real_vgg_first and not self.opt.no_vgg_loss
        with torch.cuda.stream(side):
            if vgg_first:
                # VGG(real) depends on the inputs alone (frozen weights): in front of the wait for D's update it can run
                # as soon as this stream is free, i.e. under the PREVIOUS step's backward passes
                out['y_vgg'] = self.criterionVGG.target_features(real_image)
        self._wait_d_update(side)
        with torch.cuda.stream(side):
            if with_d:
                out['pred_real'] = self.discriminate(netD_cond, real_image, mask_cond, False)
                out['loss_D_real'] = self.criterionGAN(out['pred_real'], True)
            if not self.opt.no_vgg_loss and not vgg_first:
                out['y_vgg'] = self.criterionVGG.target_features(real_image)
        # these tensors were allocated on the side stream and are read on the main one
        for feats in out.get('pred_real', ()):
            for t in feats:
                t.record_stream(main)
        if with_d:
            out['loss_D_real'].record_stream(main)
        for t in (out['y_vgg'] or []):
            t.record_stream(main)
        for t in (netD_cond, real_image, mask_cond) + tuple(getattr(netD_cond, '_him_pyramid', ())):
            t.record_stream(side)
        return out

    def _generate(self, buf, input_mask, cond_image, mask_in):
        if self.netG_type == 'global':
            return self.netG.forward(buf, mask_in)
        if self.netG_type == 'local':
            return self.netG.forward(buf)
        return self.netG.forward(cond_image, input_mask, mask_in)

    def forward(self, label, inst, image, feat, mask_in, mask_out, infer=False, obj_mask=None):
        opt = self.opt
        main = torch.cuda.current_stream(self.device)
        # The input encoding and the real-image branch touch neither network's CURRENT-step gradients nor the generator:
        # on the real-image stream they depend only on the inputs and on D's previous update.  With D updated inside
        # loss_G.backward() (optimize_parameters), step N+1's encoding + D(real) + VGG(real) run under step N's generator
        # backward, Adam and panel rebuild -- the 2.5 ms at the step boundary in which no matrix kernel ran (r04t trace).
        enc_side = None
        if self.isTrain and SCHED.real_ahead and SCHED.inputs_on_real_stream:
            enc_side = ops._real_stream(self.device)
            ready, self._inputs_ready_event = getattr(self, '_inputs_ready_event', None), None
            on_device = any(torch.is_tensor(t) and t.is_cuda for t in (label, inst, image, mask_in, mask_out, obj_mask))
            if ready is not None:
                enc_side.wait_event(ready)       # the caller's promise: the device tensors are complete at this event
            elif on_device:
                enc_side.wait_stream(main)       # device tensors of unknown origin: whatever the current stream holds
        with torch.cuda.stream(enc_side) if enc_side is not None else contextlib.nullcontext():
            input_mask, inst_map, real_image, _, cond_image = self.encode_input(label, inst, image, feat, mask_in=mask_in,
                                                                               obj_mask=obj_mask, lazy=True)
            buf, n_label, n_cond, mask_in = self._enc
            netD_cond = input_mask if self.no_imgCond else buf
            mask_cond = mask_in if not self.use_soft_mask else self._dev(mask_out)
            if self._d_split():
                # the pooled condition of the PatchGAN scales: once per step, before the streams fork (every pass reads it)
                ops.cond_pyramid(netD_cond, opt.num_D, prefill=2 if self.isTrain else 0, image_channels=opt.output_nc)
            inputs_ready = torch.cuda.Event()
            inputs_ready.record(torch.cuda.current_stream(self.device))
        if enc_side is not None:
            main.wait_event(inputs_ready)
            for t in (buf, input_mask, inst_map, real_image, cond_image, mask_in, mask_cond) + tuple(
                    getattr(netD_cond, '_him_pyramid', ())):
                if torch.is_tensor(t) or isinstance(t, ops.LabelCond):
                    t.record_stream(main)        # allocated on the real-image stream, read on the main one

        # Everything that depends only on the REAL image (its discriminator pass and its VGG features) is independent
        # of the generator: it runs on a side stream next to the generator forward and fills the matrix pipe where the
        # one-tile-per-CU ResnetBlock launches leave it idle.
        # The generator forward is enqueued FIRST: the host needs ~2.5 ms to issue the ~80 launches of the real branch,
        # and the main stream would sit idle for that long at the start of every step (r02 trace) if they went first.
        if getattr(self, '_g_update_pending', False):
            # the previous step left G's exchange + Adam + panel rebuild running on the optimizer stream: the real-image
            # branch (no generator weights) is enqueued first and runs next to them, the generator waits
            ahead = self._real_branch_ahead(netD_cond, real_image, mask_cond, inputs_ready)
            self._wait_g_update()
            fake_image = self._generate(buf, input_mask, cond_image, mask_in)
        elif SCHED.real_first:      # A/B switch: round 2's issue order
            ahead = self._real_branch_ahead(netD_cond, real_image, mask_cond, inputs_ready)
            fake_image = self._generate(buf, input_mask, cond_image, mask_in)
        else:
            fake_image = self._generate(buf, input_mask, cond_image, mask_in)
            ahead = self._real_branch_ahead(netD_cond, real_image, mask_cond, inputs_ready)
        if ahead is not None:
            from ..dist import timed_wait
            # timed (bench.py exposed_comm_ms): the real-image stream waited for D's exchange + Adam inside
            # _real_branch_ahead ('d_update_wait_real'); what of that reaches the step is bounded by this join
            timed_wait(torch.cuda.current_stream(self.device), ahead['stream'],
                       self.comm_timing['real_branch_join'] if self.comm_timing else None)
        if self.isTrain:
            self._wait_d_update()          # the previous step's D exchange + Adam (own stream) end before D is read here
            self._d_update_pending = False
        # VGG(fake) is independent of the discriminator passes on the fake image: it runs on a stream of its own next to
        # them (the small PatchGAN scales leave most of the chip idle); autograd replays its backward on that stream too.
        vgg_side = None
        if not opt.no_vgg_loss and SCHED.vgg_stream:
            main_s = torch.cuda.current_stream(self.device)
            vs = ops._vgg_stream(self.device)
            vs.wait_stream(main_s)
            fake_image.record_stream(vs)
            real_image.record_stream(vs)
            with torch.cuda.stream(vs):
                vgg_side = self.criterionVGG(fake_image, real_image, ahead['y_vgg'] if ahead is not None else None)
            vgg_side.record_stream(main_s)

        # Fake detection and loss / real detection and loss / GAN loss (:218-233).  The reference runs the discriminator
        # on the fake image twice -- once detached (loss_D_fake) and once attached (loss_G_GAN + feature matching) --
        # with identical weights and identical input values, i.e. identical activations.  With an empty image pool, and
        # when optimize_parameters() drives the step (a caller doing its own loss.backward() calls gets the reference's
        # three separate passes), the pass is computed ONCE and the gradients are routed at backward time: during
        # loss_G.backward() they
        # reach the generator but D's weight gradients are skipped (the reference computes and discards them,
        # train_mask2image.py:84); during loss_D.backward() they reach D's weights and stop in front of the generator.
        # only optimize_parameters() owns both backward calls; --sn_D: every discriminator forward moves the persisted
        # power-iteration vectors, so the reference's three passes are kept as three passes
        share = opt.pool_size == 0 and self._share_fake_pass and not self._d_stateful()
        if share:
            self._fake_gate = {'open': True}
            pred_fake = self.netD.forward(self._d_input(netD_cond, ops.grad_switch(fake_image, self._fake_gate),
                                                        mask_cond))
            pred_fake_pool = pred_fake
        else:
            self._fake_gate = None
            pred_fake_pool = self.discriminate(netD_cond, fake_image, mask_cond, True)
        loss_D_fake = self.criterionGAN(pred_fake_pool, False)
        if ahead is not None and 'pred_real' in ahead:
            pred_real, loss_D_real = ahead['pred_real'], ahead['loss_D_real']
        else:
            pred_real = self.discriminate(netD_cond, real_image, mask_cond, False)
            loss_D_real = self.criterionGAN(pred_real, True)
        if not share:
            with frozen_params():
                pred_fake = self.netD.forward(self._d_input(netD_cond, fake_image, mask_cond))
        loss_G_GAN = self.criterionGAN(pred_fake, True)

        loss_G_GAN_Feat = None
        if not opt.no_ganFeat_loss:
            feat_weights = 4.0 / (opt.n_layers_D + 1)
            D_weights = 1.0 / opt.num_D
            pairs = [(pred_fake[i][j], pred_real[i][j]) for i in range(opt.num_D) for j in range(len(pred_fake[i]) - 1)]
            # D_weights * feat_weights * L1 * lambda_feat per term (reference :235-242), folded into one weight vector
            w = float(np.float32(np.float32(D_weights * feat_weights)) * np.float32(opt.lambda_feat))
            loss_G_GAN_Feat = ops.l1_weighted_sum(pairs, [w] * len(pairs))

        loss_G_VGG = None
        if vgg_side is not None:
            torch.cuda.current_stream(self.device).wait_stream(ops._vgg_stream(self.device))
            loss_G_VGG = ops.lincomb([vgg_side], [opt.lambda_feat])
            if getattr(self, '_vgg_bwd_early', False):
                # optimize_parameters() differentiates this term on its own, from ``vgg_side`` on the VGG stream and BEFORE
                # loss_D.backward() (see there): the value stays in loss_G, the graph does not
                self._vgg_early = (vgg_side, fake_image)
                loss_G_VGG = loss_G_VGG.detach()
        elif not opt.no_vgg_loss:
            loss_G_VGG = ops.lincomb([self.criterionVGG(fake_image, real_image,
                                                        ahead['y_vgg'] if ahead is not None else None)], [opt.lambda_feat])
        if opt.lambda_rec > 0:
            rec = self.criterionFeat(fake_image, real_image)
            loss_G_GAN_Feat = ops.lincomb([rec] if loss_G_GAN_Feat is None else [loss_G_GAN_Feat, rec],
                                          [opt.lambda_rec] if loss_G_GAN_Feat is None else [1.0, opt.lambda_rec])
        if loss_G_GAN_Feat is None:
            loss_G_GAN_Feat = torch.zeros(1, device=self.device)
        if loss_G_VGG is None:
            loss_G_VGG = torch.zeros(1, device=self.device)

        # kept on the device (the reference does four blocking .cpu() copies here every step, :253-256)
        self._visuals = (fake_image.detach(), real_image, input_mask, cond_image)
        return [[loss_G_GAN, loss_G_GAN_Feat, loss_G_VGG, loss_D_real, loss_D_fake],
                None if not infer else fake_image]

    def inference(self, label, inst, image, mask_in, mask_out, obj_mask=None, color_embed=None):
        with torch.no_grad():
            input_mask, _, real_image, _, cond_image = self.encode_input(label, inst, image, mask_in=mask_in,
                                                                        infer=True, obj_mask=obj_mask,
                                                                        color_embed=color_embed, lazy=True)
            buf, _, _, mask_dev = self._enc
            self._wait_g_update()
            fake_image = self._generate(buf, input_mask, cond_image, mask_dev)
        self._visuals = (fake_image, real_image, input_mask, cond_image)
        return fake_image

    def get_edges(self, t):
        t = self._dev(t)
        B, _, H, W = t.shape
        out = torch.empty_like(t)
        from .._cabi import lib
        lib.him_edges(t.data_ptr(), out.data_ptr(), B, H, W, 1, 0, torch.cuda.current_stream().cuda_stream)
        return out

    def get_current_visuals(self):
        fake, real, label, cond = self._visuals
        if isinstance(label, ops.LabelCond):
            label = label.full()
        return OrderedDict([('input_label', label[0].cpu()), ('input_image', cond[0].cpu()),
                            ('real_image', real[0].cpu()), ('synthesized_image', fake[0].cpu())])

    # ------------------------------------------------------------------------------------------
    # the optimisation step: train_mask2image.py:68-86
    # ------------------------------------------------------------------------------------------
    def combine_losses(self, losses):
        # torch.mean over the (DataParallel) replica axis: one process per GPU holds ONE value per loss -- a view, no kernel
        losses = [(x.reshape(()) if x.numel() == 1 else torch.mean(x)) if not isinstance(x, int) else x for x in losses]
        loss_dict = dict(zip(self.loss_names, losses))
        # (D_fake + D_real) * 0.5 and G_GAN + G_GAN_Feat + G_VGG (train_mask2image.py:70-71), one launch each
        self.loss_D = ops.lincomb([loss_dict['D_fake'], loss_dict['D_real']], scale=0.5)
        self.loss_G = ops.lincomb([loss_dict['G_GAN'], loss_dict['G_GAN_Feat'], loss_dict['G_VGG']])
        return loss_dict

    def _run_backward_G(self, last=False, extra_root=None):
        """loss_G.backward() with the shared fake-image D pass routed to the generator only.  ``last``: loss_D.backward()
        has already run (the shared graph may be freed).  ``extra_root``: (fake_image, gradient) of a loss term that was
        differentiated down to the fake image beforehand."""
        shared = self._fake_gate is not None
        if shared:
            self._fake_gate['open'] = True
            ops.SKIP_WGRAD.update(self._d_weight_ids)
        try:
            if extra_root is not None:
                torch.autograd.backward([self.loss_G, extra_root[0]], [None, extra_root[1]],
                                        retain_graph=shared and not last)
            else:
                self.loss_G.backward(retain_graph=shared and not last)
        finally:
            ops.SKIP_WGRAD.difference_update(self._d_weight_ids)

    def _run_backward_D(self, first=False):
        """loss_D.backward() with the shared fake-image D pass routed to D's weights only.  ``first``: loss_G.backward()
        comes afterwards (keep the shared graph; the VGG stream is busy then, so D's weight gradients take the ordinary
        weight-gradient stream, idle until the generator's backward starts)."""
        shared = self._fake_gate is not None
        if shared:
            self._fake_gate['open'] = False
            ops.SKIP_DGRAD.update(self._d_first_weight_ids)
        # D's weight gradients -- fake branch issued from the main stream, real branch from the real-image stream where
        # its forward ran -- all go to the VGG stream (idle by now) instead of queueing behind the generator's last weight
        # gradients on the side stream.  ONE stream for both branches: they accumulate into the same arena slots.
        main = torch.cuda.current_stream(self.device)
        wg = ops._vgg_stream(self.device)
        routes = {main: wg, ops._real_stream(self.device): wg}
        try:
            with ops.route_wgrads(routes if (SCHED.d_wgrad_routes and not first) else {}):
                self.loss_D.backward(retain_graph=shared and first)
        finally:
            ops.SKIP_DGRAD.difference_update(self._d_first_weight_ids)
        # the real branch's data-gradient chain reads D's weights on its own stream: D's Adam step comes after it
        main.wait_stream(ops._real_stream(self.device))

    def backward_G(self):
        """optimizer_G.zero_grad(); loss_G.backward(); optimizer_G.step()   (:78-80)."""
        self.optimizer_G.zero_grad()
        if self.reducer_G is not None:
            self.reducer_G.begin()
        self._run_backward_G()
        if self.reducer_G is not None:
            self.reducer_G.finish()
        self.optimizer_G.step()

    def backward_D(self):
        """optimizer_D.zero_grad(); loss_D.backward(); optimizer_D.step()   (:84-86)."""
        self.optimizer_D.zero_grad()
        if self.reducer_D is not None:
            self.reducer_D.begin(contributions=2)
        self._run_backward_D()
        if self.reducer_D is not None:
            self.reducer_D.finish()
        self.optimizer_D.step()

    def optimize_parameters(self, data=None, infer=False):
        """One full training step on a batch dict (keys of SegmentationDataset: label, inst, image, mask_in,
        mask_out[, obj_mask]).  Same arithmetic as backward_G(); backward_D(), in the shipped order
        (``SCHED.d_backward_first``): both arenas zeroed, ``loss_D.backward()`` FIRST (its graph holds no generator
        parameter: the fake is detached / gated) with D's 34 MB exchange going out under it, then ``loss_G.backward()`` --
        the LAST thing in the step -- with G's 730 MB leaving bucket by bucket under the rest of its own backward; D's Adam
        waits for the generator's backward (loss_G differentiates through D's current weights); G's last buckets + Adam +
        panel rebuild are left running on the optimizer stream into the NEXT step's input encoding and real-image branch
        (``_wait_g_update``).  The exchange budget of this order is written down in DESIGN.md 6.  With
        ``d_backward_first`` off (``bench.py --g-backward-first``) the reference's order runs: G's exchange + Adam then
        hide under the whole of D's backward."""
        data = data if data is not None else self.input
        kw = dict(label=data['label'], inst=data['inst'], image=data['image'], feat=None, mask_in=data['mask_in'],
                  mask_out=data['mask_out'], infer=infer)
        if 'obj_mask' in data:
            kw['obj_mask'] = data['obj_mask']
        self._share_fake_pass = SCHED.share_fake_pass
        # device-resident batches may carry 'ready_event' (torch.cuda.Event recorded behind the kernels / copies that
        # produced them): the input encoding then waits for THAT instead of for the whole current stream (forward())
        self._inputs_ready_event = data.get('ready_event') if hasattr(data, 'get') else None
        gan = not self.opt.no_gan
        zero_ev = None
        if SCHED.zero_grad_side and SCHED.wgrad_stream and self.reducer_G is None and self.reducer_D is None:
            # both arenas zeroed on the weight-gradient stream NOW, under the forward pass (nothing reads the gradients until
            # the backward pass; the weight-gradient kernels follow on the same stream) -- the fill used to sit on the main
            # stream between the losses and the first backward kernel (one rank: -0.37 ms per step).  NOT with a gradient
            # exchange attached: the fill makes the weight-gradient stream a FIFTH busy hardware queue at the step boundary,
            # next to the exchange + Adam on the optimizer streams -- measured with the exchange stand-in: +2.2 ms per step
            # (profiles/r05_ab_log.txt; the queue cliff of DESIGN.md 3); on the optimizer streams behind Adam it gains nothing
            main0 = torch.cuda.current_stream(self.device)
            ws = ops._side_stream(self.device)
            ws.wait_stream(main0)             # the caller may have read / written .grad on the current stream
            ws.wait_stream(ops._opt_stream(self.device))      # the previous step's Adam kernels read them
            ws.wait_stream(ops._d_opt_stream(self.device))
            with torch.cuda.stream(ws):
                self.optimizer_G.zero_grad()
                if gan:
                    self.optimizer_D.zero_grad()
                zero_ev = torch.cuda.Event()
                zero_ev.record(ws)
        self._vgg_bwd_early = SCHED.vgg_backward_early and SCHED.d_backward_first and not self.opt.no_gan
        self._vgg_early = None
        try:
            losses, generated = self.forward(**kw)
        finally:
            self._share_fake_pass = False
            self._vgg_bwd_early = False
        loss_dict = self.combine_losses(losses)
        # Same arithmetic as backward_G(); backward_D(), reordered: both arenas are zeroed first, the generator's Adam
        # step (and its all-reduce) is deferred behind loss_D.backward() -- legal because loss_D's graph holds no
        # generator parameter (the fake is detached / gated) -- so the 730 MB gradient exchange and G's Adam step hide
        # under D's backward.
        main = torch.cuda.current_stream(self.device)
        if zero_ev is None:
            self.optimizer_G.zero_grad()
            if gan:
                self.optimizer_D.zero_grad()
        else:
            main.wait_event(zero_ev)          # autograd's own accumulations / routed weight gradients follow the fill
            ops._vgg_stream(self.device).wait_event(zero_ev)      # D's weight gradients may be routed there
        opt_stream = ops._opt_stream(self.device)
        from ..dist import timed_wait
        if gan and SCHED.d_backward_first:
            # loss_D.backward() FIRST.  Its graph hangs off the discriminator passes only (the fake is detached / gated), so
            # it can start as soon as those are enqueued: its data-gradient chains (main stream / real-image stream) and
            # weight gradients (weight-gradient stream) then fill the window in which the main stream otherwise waits for
            # VGG(fake) forward + backward, instead of forming a 12 ms tail behind the generator's backward (r03b trace).
            # D's Adam waits for the generator's backward: loss_G still differentiates THROUGH D's current weights.
            # (HIM_VGG_BACKWARD_EARLY=1, measured slower) The VGG term of loss_G first, down to the fake image and no
            # further: its backward lives on the VGG stream (autograd replays a node on the stream of its forward) but, as
            # part of loss_G.backward(), can only start once the MAIN stream has worked through loss_D's backward and reached
            # loss_G's root -- the r03 trace shows the fused-Winograd kernels of VGG's backward running alone for 5.4 ms with
            # the main stream idle behind them.  Started here it runs next to loss_D.backward(); the generator's backward
            # then starts from d(GAN + feature matching)/d(fake) + this gradient.
            early = None
            if self._vgg_early is not None:
                vgg_side, fake_image = self._vgg_early
                self._vgg_early = None
                vs = ops._vgg_stream(self.device)
                with torch.cuda.stream(vs):
                    seed = torch.full_like(vgg_side, float(self.opt.lambda_feat))
                    (g_vgg,) = torch.autograd.grad(vgg_side, fake_image, grad_outputs=seed)
                g_vgg.record_stream(main)
                early = (fake_image, g_vgg)
            if self.reducer_D is not None:
                self.reducer_D.begin(contributions=2)
            self._run_backward_D(first=True)
            if self.reducer_G is not None:
                self._g_chunked = self.reducer_G.bucket_hook is not None and (
                    SCHED.adam_chunked or (SCHED.adam_chunked_dp and not getattr(self.reducer_G, 'local', False)))
                if self._g_chunked:
                    self.optimizer_G.begin_step()
                self.reducer_G.begin()
            if early is not None:
                main.wait_stream(ops._vgg_stream(self.device))
            # D's exchange + Adam: from INSIDE loss_G.backward(), the moment the gradient has passed back through the
            # discriminator to the fake image (ops._GradSwitch.backward, autograd's thread) -- nothing reads D's weights
            # after that, so the update (and with it the next step's real-image branch, which waits for nothing else) no
            # longer queues behind the generator's whole backward.  Without a shared fake pass: after the backward, as before.
            # One rank only: next to a gradient exchange the early update measures the same step (52.33 vs 52.28 ms with the
            # exchange stand-in, profiles/r05_ab_log.txt) and would move D's collectives in front of G's in issue order --
            # data-parallel ranks keep the order every multi-rank test has run
            d_early = SCHED.d_update_early and self.reducer_G is None and self.reducer_D is None
            if d_early and self._fake_gate is not None:
                self._fake_gate['on_open_backward'] = self._d_update
            ops.take_stem_pre(self.device)      # a stale record of an earlier backward (backward_G(), another model)
            self._run_backward_G(last=True, extra_root=early)
            if self._fake_gate is None or self._fake_gate.pop('on_open_backward', None) is not None or not d_early:
                self._d_update()
            # GlobalGenerator is a chain: its stem's backward is the last node, and the stem's run-length weight gradient
            # (0.5 ms, LDS-bound, one workgroup per CU) the last kernel of the pass.  Everything ELSE is final one kernel
            # earlier: that part of Adam (5 GB of HBM traffic, no LDS) + the panel rebuild start next to it, behind the two
            # events ops recorded in front of the stem's weight gradient; step() below closes with the stem's slice.
            pre = ops.take_stem_pre(self.device)
            if (pre is not None and SCHED.adam_split_stem and self.netG_type == 'global' and self.reducer_G is None
                    and pre[3] == id(self.netG.model[1].weight)):
                ev_main, ev_side, (lo, hi), _ = pre
                hi = min((hi + 63) // 64 * 64, self.optimizer_G.arena.total)     # whole 256-byte slots (padding: zero gradient)
                with torch.cuda.stream(opt_stream):
                    opt_stream.wait_event(ev_main)
                    opt_stream.wait_event(ev_side)
                    for ev in ops.wgrad_waitables(self.device)[1]:     # weight gradients routed to other streams (all issued
                        opt_stream.wait_event(ev)                      # before the stem's backward, the last node)
                    self.optimizer_G.begin_step()
                    try:
                        if lo > 0:
                            self.optimizer_G.step_range(0, lo)
                        if hi < self.optimizer_G.arena.total:
                            self.optimizer_G.step_range(hi, self.optimizer_G.arena.total)
                    except Exception:
                        self.optimizer_G.abort_step()
                        raise
            # G's exchange + Adam + panel rebuild (5 GB of HBM traffic, no matrix work) on their own stream, NOT waited for
            # here: the next step's input encoding and real-image branch do not touch G and run next to them; the next
            # generator forward waits (forward()); anything else that reads parameters calls sync() first.
            opt_stream.wait_stream(main)
            with torch.cuda.stream(opt_stream):
                if self.reducer_G is not None:
                    self.reducer_G.finish()
                self.optimizer_G.step()
                # what the next generator forward waits for (NOT the whole stream: the next step's arena fill follows here)
                self._g_update_done = torch.cuda.Event(enable_timing=False)
                self._g_update_done.record(opt_stream)
            self._g_chunked = False
            self._g_update_pending = True
        else:
            if self.reducer_G is not None:
                self._g_chunked = self.reducer_G.bucket_hook is not None and (
                    SCHED.adam_chunked or (SCHED.adam_chunked_dp and not getattr(self.reducer_G, 'local', False)))
                if self._g_chunked:
                    self.optimizer_G.begin_step()
                self.reducer_G.begin()
            self._run_backward_G()
            # G's exchange + Adam step go to their own stream: they wait for G's data-gradient chain (main) and weight
            # gradients (side stream), then run under D's backward
            opt_stream.wait_stream(main)
            with torch.cuda.stream(opt_stream):
                if self.reducer_G is not None:
                    self.reducer_G.finish()
                self.optimizer_G.step()
            self._g_chunked = False
            if gan:
                if self.reducer_D is not None:
                    self.reducer_D.begin(contributions=2)
                self._run_backward_D()
            timed_wait(main, opt_stream, self.comm_timing['g_update_tail'] if self.comm_timing else None)
            if gan:
                if self.reducer_D is not None:
                    # D's exchange (34 MB over xGMI) + Adam step go to a stream of their own and are NOT waited for here:
                    # D's parameters are first needed by the next step's discriminator passes, so the exchange hides under
                    # the next encode_input + generator forward (forward() makes the consumers wait, see _wait_d_update)
                    d_stream = ops._d_opt_stream(self.device)
                    d_stream.wait_stream(main)
                    with torch.cuda.stream(d_stream):
                        self.reducer_D.finish()
                        self.optimizer_D.step()
                    self._d_update_pending = True
                else:
                    self.optimizer_D.step()
        self.generated = generated
        # both graphs have been consumed: drop them now (not at the next forward), so the previous step's G+D+VGG
        # activations are not resident while the next forward allocates its own
        self.loss_G = self.loss_D = None
        self._fake_gate = None
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss_dict.items()}

    def _bucket_update_G(self, b, start, end):
        """dist.GradReducer.bucket_hook of the generator: gradient bucket [start, end) is final (all its weight gradients
        queued, exchanged when data parallel) and the reducer has made its stream -- the generator's optimizer stream -- wait
        for them: the bucket's Adam step + panel rebuild go right behind, under the rest of the backward pass.  Only inside
        optimize_parameters() (``_g_chunked``): nothing reads a layer's weights after its own backward there."""
        if self._g_chunked:
            self.optimizer_G.step_range(start, end)

    def _d_update(self):
        """D's gradient exchange (data parallel) + Adam step + panel rebuild on D's optimizer stream, after everything the
        CALLING thread's current stream holds (the discriminator's data gradients) and D's weight gradients."""
        d_stream = ops._d_opt_stream(self.device)
        d_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(d_stream):
            if self.reducer_D is not None:
                self.reducer_D.finish()
            self.optimizer_D.step()
        self._d_update_pending = True

    def _wait_g_update(self):
        """Make the current stream wait for a generator update still running on the optimizer stream."""
        if getattr(self, '_g_update_pending', False):
            from ..dist import timed_wait
            cur = torch.cuda.current_stream(self.device)
            ev = getattr(self.optimizer_G, 'updated', None) if SCHED.panel_pipeline else None
            if ev is None:
                ev, self._g_update_done = getattr(self, '_g_update_done', None), None    # Adam + the whole panel rebuild
            if ev is None:
                timed_wait(cur, ops._opt_stream(self.device), self.comm_timing['g_update_tail'] if self.comm_timing else None)
            else:
                # wait for the Adam kernel only: the weight panels are rebuilt behind it in forward order and every conv
                # waits for ITS panel's event (ops._panel), so the forward runs down the net behind the rebuild pass
                sink = self.comm_timing['g_update_tail'] if self.comm_timing else None
                if sink is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(cur)
                    cur.wait_event(ev)
                    e1.record(cur)
                    sink.append((e0, e1))
                else:
                    cur.wait_event(ev)
            self._g_update_pending = False

    def _wait_d_update(self, stream=None):
        """Make ``stream`` (default: the current one) wait for a discriminator update still running on its own stream."""
        if getattr(self, '_d_update_pending', False):
            from ..dist import timed_wait
            key = 'd_update_wait' if stream is None else 'd_update_wait_real'
            timed_wait(stream or torch.cuda.current_stream(self.device), ops._d_opt_stream(self.device),
                       self.comm_timing[key] if self.comm_timing else None)

    def start_comm_timing(self):
        """From now on every wait of a compute stream for the gradient exchange / a deferred optimizer step is bracketed
        by timing events (dist.timed_wait).  ``read_comm_timing`` turns them into milliseconds per step."""
        self.comm_timing = {'g_update_tail': [], 'd_update_wait': [], 'g_exchange_wait': [], 'd_exchange_wait': [],
                            'd_update_wait_real': [], 'real_branch_join': []}
        if self.reducer_G is not None:
            self.reducer_G.timing = self.comm_timing['g_exchange_wait']
        if self.reducer_D is not None:
            self.reducer_D.timing = self.comm_timing['d_exchange_wait']

    def read_comm_timing(self, steps):
        """ms per step a stream sat idle: ``g_exchange_wait`` = the optimizer stream waiting for G's all-reduce after
        the last weight gradient; ``g_update_tail`` = the main stream, after loss_D.backward(), waiting for G's exchange +
        Adam; ``d_exchange_wait`` = D's update stream waiting for D's all-reduce; ``d_update_wait`` = the NEXT step's
        discriminator pass waiting for D's exchange + Adam on the MAIN stream -- normally ~0, because the real-image
        stream has already waited for it: ``d_update_wait_real`` = that stream's idle time, of which at most
        ``real_branch_join`` (the main stream waiting for the real-image branch, compute included) reaches the step."""
        torch.cuda.synchronize(self.device)
        out = {k: round(sum(a.elapsed_time(b) for a, b in v) / max(steps, 1), 4) for k, v in self.comm_timing.items()}
        self.comm_timing = None
        for r in (self.reducer_G, self.reducer_D):
            if r is not None:
                r.timing = None
        return out

    def sync(self):
        """Join every helper stream into the current one (before parameters are read from outside the step)."""
        cur = torch.cuda.current_stream(self.device)
        ops.join_side_stream(self.device)
        cur.wait_stream(ops._opt_stream(self.device))
        cur.wait_stream(ops._real_stream(self.device))
        self._g_update_pending = False
        self._wait_d_update(cur)
        self._d_update_pending = False

    # ------------------------------------------------------------------------------------------
    def save(self, which_epoch):
        self.sync()
        self.save_network(self.netG, 'G', which_epoch, self.gpu_ids)
        self.save_network(self.netD, 'D', which_epoch, self.gpu_ids)

    def delete_model(self, which_epoch):
        self.delete_network('G', which_epoch, self.gpu_ids)
        self.delete_network('D', which_epoch, self.gpu_ids)

    def update_fixed_params(self):
        self.sync()
        self.optimizer_G = FusedAdam(self.netG.parameters(), lr=self.opt.lr, betas=(self.opt.beta1, 0.999),
                                     arena=self.optimizer_G.arena)
        print('------------ Now also finetuning global generator -----------')

    def update_learning_rate(self):
        lrd = self.opt.lr / self.opt.niter_decay
        lr = self.old_lr - lrd
        for param_group in self.optimizer_D.param_groups:
            param_group['lr'] = lr
        for param_group in self.optimizer_G.param_groups:
            param_group['lr'] = lr
        if getattr(self.opt, 'verbose', False):
            print('update learning rate: %f -> %f' % (self.old_lr, lr))
        self.old_lr = lr
