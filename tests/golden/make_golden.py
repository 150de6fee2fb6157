"""Generate the golden vectors under tests/golden/ by running the REAL reference (imported from
/root/reference through oracle/ref_shim.py) and, in the same breath, assert that oracle/ref_cpu.py
reproduces it.  Runs only in the build container.  Usage:

    python tests/golden/make_golden.py [tiny] [flags] [c1] [c2]

Fixtures hold arrays + the flag dict only (no reference source).  Weights/batches are NOT stored: they
are regenerated from neurips18_hierchical_image_manipulation_amd.synth seeds (G=1, D=2, VGG=3; batch
seed = 1000+step*64+rank).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from oracle import ref_shim, ref_cpu                                     # noqa: E402
from neurips18_hierchical_image_manipulation_amd import synth            # noqa: E402

SEED_G, SEED_D, SEED_V = 1, 2, 3
NAMES = ref_cpu.Mask2ImageModel.loss_names


def flags_to_argv(fl):
    argv = ['--name', 'g', '--checkpoints_dir', '/tmp/him_golden_ck', '--gpu_ids', '0']
    for k, v in fl.items():
        if isinstance(v, bool):
            if v:
                argv.append('--' + k)
        else:
            argv += ['--' + k, str(v)]
    return argv


def load_same(rm, om):
    sdG = synth.init_state_dict(om.netG.state_dict(), SEED_G)
    sdD = synth.init_state_dict(om.netD.state_dict(), SEED_D)
    assert list(rm.netG.state_dict().keys()) == list(sdG.keys())
    assert list(rm.netD.state_dict().keys()) == list(sdD.keys())
    for m in (rm, om):
        m.netG.load_state_dict(sdG)
        m.netD.load_state_dict(sdD)
    if om.vgg is not None:
        sdV = synth.init_state_dict(om.vgg.state_dict(), SEED_V, 'vgg')
        rm.criterionVGG.vgg.load_state_dict(sdV)
        om.vgg.load_state_dict(sdV)


def trajectory(tag, fl, B, H, W, steps, color=False, save_outputs=False, save_keys=False):
    t0 = time.time()
    rm, _ = ref_shim.make_model(flags_to_argv(dict(fl, batchSize=B)), color=color)
    om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**fl))
    load_same(rm, om)
    extra = {}
    if save_keys:       # the checkpoint ABI of this flag set: the REFERENCE's state-dict key names, in its order
        extra['g_keys'] = np.array(list(rm.netG.state_dict().keys()))
        extra['d_keys'] = np.array(list(rm.netD.state_dict().keys()))
    if save_outputs:
        b0 = synth.make_batch(0, 0, B, H, W, fl.get('label_nc', 35), color)
        with torch.no_grad():
            onehot, cond = om.encode_input(b0['label'], b0['inst'], b0['image'], b0['mask_in'],
                                           b0.get('obj_mask'))
            if color:
                r_in = rm.encode_input(b0['label'], b0['inst'], b0['image'], None, mask_in=b0['mask_in'],
                                       obj_mask=b0['obj_mask'])
            else:
                r_in = rm.encode_input(b0['label'], b0['inst'], b0['image'], None, mask_in=b0['mask_in'])
            assert torch.equal(r_in[0], onehot) and torch.allclose(r_in[4], cond, atol=1e-7)
            fake_o = om.generate(onehot, cond, b0['mask_in'])
            if fl['netG'] == 'global':
                fake_r = rm.netG.forward(torch.cat((r_in[0], r_in[4]), 1), b0['mask_in'])
            else:
                fake_r = rm.netG.forward(r_in[4], r_in[0], b0['mask_in'])
            assert torch.allclose(fake_r, fake_o, atol=1e-6), (fake_r - fake_o).abs().max()
            d_in = torch.randn(B, om.d_in, H, W, generator=torch.Generator().manual_seed(5))
            dr, do = rm.netD(d_in), om.netD(d_in)
            for a, b in zip(dr, do):
                for x, y in zip(a, b):
                    assert torch.allclose(x, y, atol=1e-6)
            extra['fake0'] = fake_r.numpy()
            extra['cond0'] = r_in[4].numpy()
            extra['d_in'] = d_in.numpy()
            for i, sc in enumerate(dr):
                extra['d_logits%d' % i] = sc[-1].numpy()
                extra['d_feat%d_0' % i] = sc[0].numpy()
    ref_l, ora_l = [], []
    uni = torch.Tensor.uniform_
    if color:  # colour noise U(0.97,1.03) -> exactly 1.0 in parity mode
        torch.Tensor.uniform_ = lambda self, *a, **k: self.fill_(0.5)
    try:
        for s in range(steps):
            b = synth.make_batch(s, 0, B, H, W, fl.get('label_nc', 35), color)
            r = ref_shim.ref_step(rm, b, color)
            o = om.optimize_parameters(b)
            ref_l.append([r[k] for k in NAMES])
            ora_l.append([o[k] for k in NAMES])
    finally:
        torch.Tensor.uniform_ = uni
    if fl.get('norm') == 'batch':
        # BatchNorm bookkeeping after the last step: the running statistics and forward counts of the first BatchNorm of
        # each net (the reference runs the discriminator three times per step: fake detached, real, fake)
        for tagn, rn, on in (('g', rm.netG, om.netG), ('d', rm.netD, om.netD)):
            rs, os_ = rn.state_dict(), on.state_dict()
            k = [k for k in rs if k.endswith('running_mean')][0]
            for kk in (k, k[:-4] + 'var', k[:-12] + 'num_batches_tracked'):
                assert torch.equal(rs[kk], os_[kk]), kk
            extra['bn_%s_key' % tagn] = np.array(k)
            extra['bn_%s_running_mean' % tagn] = rs[k].numpy()
            extra['bn_%s_running_var' % tagn] = rs[k[:-4] + 'var'].numpy()
            extra['bn_%s_batches' % tagn] = rs[k[:-12] + 'num_batches_tracked'].numpy()
    ref_l, ora_l = np.array(ref_l, np.float64), np.array(ora_l, np.float64)
    rel = np.abs(ref_l - ora_l) / np.maximum(np.abs(ref_l), 1e-12)
    print('%s: %d steps, max rel(oracle vs reference) = %.3e, %.1fs' % (tag, steps, rel.max(), time.time() - t0))
    assert rel.max() < 1e-5, rel
    np.savez_compressed(os.path.join(HERE, tag + '.npz'), flags=json.dumps(fl), B=B, H=H, W=W,
                        color=int(color), losses=ref_l.astype(np.float32), loss_names=np.array(NAMES), **extra)


def golden_nets():
    """Forward of classes the reference models cannot reach (LocalEnhancer) + SN + edges."""
    P, D, L, S, U = ref_shim.nets()
    out = {}
    # LocalEnhancer (models/Pix2Pix_NET.py:8-61)
    r = P.LocalEnhancer(9, 3, ngf=4, n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2)
    o = ref_cpu.LocalEnhancer(9, 3, ngf=4, n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2)
    sd = synth.init_state_dict(o.state_dict(), 11)
    assert list(r.state_dict().keys()) == list(sd.keys())
    r.load_state_dict(sd)
    o.load_state_dict(sd)
    x = torch.randn(2, 9, 32, 64, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        yr, yo = r(x), o(x)
    assert torch.allclose(yr, yo, atol=1e-6)
    out.update(local_x=x.numpy(), local_y=yr.numpy())
    # the same class with --norm batch (BatchNorm2d(affine=True), training mode: batch statistics), round 6
    r = P.LocalEnhancer(9, 3, ngf=4, n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2,
                        norm_layer='batch')
    o = ref_cpu.LocalEnhancer(9, 3, ngf=4, n_downsample_global=2, n_blocks_global=2, n_local_enhancers=1, n_blocks_local=2,
                              norm_layer='batch')
    sd = synth.init_state_dict(o.state_dict(), 12)
    assert list(r.state_dict().keys()) == list(sd.keys())
    r.load_state_dict(sd)
    o.load_state_dict(sd)
    with torch.no_grad():
        yr, yo = r(x), o(x)
    assert torch.equal(yr, yo)
    assert all(torch.equal(a, b) for a, b in zip(r.state_dict().values(), o.state_dict().values()))   # running statistics
    out.update(local_bn_y=yr.numpy(), local_bn_keys=np.array(list(sd.keys())))
    # spectral norm (models/sn_utils.py:8-25): sigma, u', W/sigma, d sigma / dW
    for tag, shape in (('sn_small', (16, 8, 3, 3)), ('sn_big', (512, 256, 4, 4))):
        g = torch.Generator().manual_seed(13)
        W = (torch.randn(*shape, generator=g) * 0.05).requires_grad_(True)
        u = torch.randn(1, shape[0], generator=g)
        sig_r, u_r = S.max_singular_value(W, u, 1)
        (gW_r,) = torch.autograd.grad(sig_r.sum(), W)
        W2 = W.detach().clone().requires_grad_(True)
        sig_o, u_o = ref_cpu.max_singular_value(W2, u, 1)
        (gW_o,) = torch.autograd.grad(sig_o.sum(), W2)
        assert torch.allclose(sig_r, sig_o) and torch.allclose(gW_r, gW_o, atol=1e-7)
        out[tag + '_sigma'] = sig_r.detach().numpy()
        out[tag + '_u'] = u_r.detach().numpy()
        if tag == 'sn_small':
            out[tag + '_W'] = W.detach().numpy()
            out[tag + '_u0'] = u.numpy()
            out[tag + '_gW'] = gW_r.numpy()
        else:
            out[tag + '_gW_sum'] = np.array([gW_r.double().sum().item(), gW_r.double().abs().sum().item()])
    # instance edges (models/pix2pixHD_condImg_model.py:285-291)
    rm, _ = ref_shim.make_model(flags_to_argv(dict(model='pix2pixHD_condImg', netG='global', ngf=4, ndf=4,
                                                   n_blocks_global=1, num_D=1, label_nc=35, no_vgg_loss=True)))
    inst = torch.from_numpy(np.random.Generator(np.random.Philox(5)).integers(0, 3, (2, 1, 16, 32)).astype(np.float32))
    er = rm.get_edges(inst)
    assert torch.equal(er, ref_cpu.get_edges(inst))
    out.update(edge_inst=inst.numpy(), edge_map=er.numpy())
    np.savez_compressed(os.path.join(HERE, 'nets_misc.npz'), **out)
    print('nets_misc: LocalEnhancer / SN / edges pinned')


def golden_box2mask_net():
    """Second hot path: the box2mask generator MaskTwoStreamConvSwitch_NET with the flags of
    scripts/train_box2mask_city.sh, 64x64 inputs, batch 2.  Forward in training and eval mode is pinned bit-exactly
    (oracle vs imported reference); the reference's parameter gradients are obtained under
    torch.autograd.graph.allow_mutation_on_saved_tensors (its in-place ReLUs alias saved tensors) and pinned through
    per-parameter sums; BatchNorm running statistics after one training-mode forward are stored for two layers."""
    from oracle import ref_mask_cpu
    ref = ref_shim.box2mask_generator()
    ora = ref_mask_cpu.MaskTwoStreamConvSwitchNet()
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys())
    sd = synth.init_state_dict(ora.state_dict(), 21)
    x = torch.randn(2, 70, 64, 64, generator=torch.Generator().manual_seed(3))
    gy = [torch.randn(2, 35, 64, 64, generator=torch.Generator().manual_seed(5)),
          torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(6))]
    # inputs are regenerated from the seeds by the tests; the checksums guard against a generator change
    out = dict(x_sum=np.array([x.double().sum().item(), x.double().abs().sum().item()]),
               gy_sum=np.array([gy[0].double().sum().item(), gy[1].double().sum().item()]))
    for mode in ('eval', 'train'):
        ref.load_state_dict(sd)
        ora.load_state_dict(sd)
        getattr(ref, mode)()
        getattr(ora, mode)()
        with torch.autograd.graph.allow_mutation_on_saved_tensors():
            a = ref(x.clone(), None)
            ((a[1] * gy[0]).sum() + (a[3] * gy[1]).sum()).backward()
        b = ora(x.clone())
        ((b[1] * gy[0]).sum() + (b[3] * gy[1]).sum()).backward()
        for p, q in zip(a, b):
            assert torch.equal(p, q), (mode, (p - q).abs().max())
        out['ctx_prob_' + mode] = a[1].detach().numpy()
        out['obj_prob_' + mode] = a[3].detach().numpy()
        names, sums = [], []
        gr = dict(ref.named_parameters())
        for k, po in ora.named_parameters():
            gr_k, go_k = gr[k].grad, po.grad
            scale = float(gr_k.abs().max())
            if k.endswith('bias'):   # conv biases in front of a BatchNorm have a TRUE gradient of 0: what is left is
                scale = max(scale, float(gr[k[:-4] + 'weight'].grad.abs().max()))   # rounding noise of the layer's scale
            assert float((gr_k - go_k).abs().max()) <= 4e-6 * max(scale, 1e-3), (mode, k, scale)
            names.append(k)
            sums.append([gr_k.double().sum().item(), gr_k.double().abs().sum().item()])
        for k, po in ora.named_parameters():
            gr[k].grad = None
            po.grad = None
        out['grad_names'] = np.array(names)
        out['grad_sums_' + mode] = np.array(sums)
        if mode == 'train':
            st = ref.state_dict()
            for k in ('conv_encoder_modules.1.running_mean', 'conv_encoder_modules.1.running_var',
                      'ctx_conv_decoder_modules.3.deep.2.running_mean', 'ctx_conv_decoder_modules.3.deep.2.running_var'):
                assert torch.equal(st[k], ora.state_dict()[k])
                out['after_' + k.replace('.', '_')] = st[k].numpy()
    np.savez_compressed(os.path.join(HERE, 'box2mask_net.npz'), **out)
    print('box2mask_net: forward (train/eval) bit-exact, gradients and running statistics pinned')


def golden_box2mask_traj(steps=6):
    """box2mask training step (TwoStreamAE_mask.forward = losses + G Adam + D Adam) of the REAL reference, run under
    torch.autograd.graph.allow_mutation_on_saved_tensors, next to the oracle restatement: 6 steps, 64x64, batch 2,
    ndf 16, the remaining flags of scripts/train_box2mask_city.sh."""
    from oracle import ref_mask_cpu
    fl = dict(ndf=16, label_nc=35, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999)
    ref = ref_shim.box2mask_trainer(**fl)
    ora = ref_mask_cpu.TwoStreamAEMask(label_nc=35, ndf=16, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999)
    assert list(ref.netD.state_dict().keys()) == list(ora.netD.state_dict().keys())
    sdG = synth.init_state_dict(ora.netG.state_dict(), 21)
    sdD = synth.init_state_dict(ora.netD.state_dict(), 22)
    for m in (ref, ora):
        m.netG.load_state_dict(sdG)
        m.netD.load_state_dict(sdD)
    rl, ol = [], []
    for s in range(steps):
        b = synth.make_box2mask_batch(s, 0, 2, 64, 64, 35)
        with torch.autograd.graph.allow_mutation_on_saved_tensors():
            r, _ = ref.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'],
                               b['mask_in'], eval_mode=False)
        rl.append([float(x.detach().reshape(-1)[0]) if torch.is_tensor(x) else float(x) for x in r])
        o = ora.step(b)
        ol.append([o[k] for k in ref_mask_cpu.LOSS_NAMES])
    rl, ol = np.array(rl, np.float64), np.array(ol, np.float64)
    rel = np.abs(rl - ol) / np.maximum(np.abs(rl), 1e-12)
    print('box2mask_traj: %d steps, max rel(oracle vs reference) per step = %s' % (
        steps, ' '.join('%.1e' % v for v in rel.max(1))))
    # free-running GAN + BatchNorm training amplifies the rounding difference between the two runs step by step (same
    # mechanism as tests/golden/chaos_envelope.json for mask2image): exact at step 0, 1e-7 for the first steps
    assert rel[:3].max() < 2e-6 and rel.max() < 5e-3, rel
    np.savez_compressed(os.path.join(HERE, 'box2mask_traj.npz'), flags=json.dumps(fl), B=2, H=64, W=64,
                        losses=rl.astype(np.float32), loss_names=np.array(ref_mask_cpu.LOSS_NAMES))


# round 6: the box2mask flags the HIP path used to refuse, each on the box2mask_traj configuration (64x64, batch 2, ndf 16):
# the parser's default generator (no --no_comb: MaskTwoStreamConv_NET, object-gated combination of the two streams'
# logits), --objReconLoss l1 | none, --which_gan patch (one PatchGAN with a Sigmoid + BCE on the LAST sample's map,
# losses.py:50-53), --which_stream obj | context, --cond_in ctx | obj, --use_simpleRes, and two mixes
BOX2MASK_VARIANTS = {
    'b2m_comb': dict(no_comb=False),
    'b2m_obj_l1': dict(objReconLoss='l1'),
    'b2m_obj_none': dict(objReconLoss='none'),
    'b2m_gan_patch': dict(which_gan='patch'),
    'b2m_gan_patch_res': dict(which_gan='patch_res'),
    'b2m_stream_obj': dict(which_stream='obj'),
    'b2m_stream_context': dict(which_stream='context'),
    'b2m_cond_ctx': dict(cond_in='ctx'),
    'b2m_cond_obj': dict(cond_in='obj'),
    'b2m_simple_res': dict(use_simpleRes=True),
    'b2m_comb_simple_nogate_instance': dict(no_comb=False, use_simpleRes=True, use_output_gate=False,
                                            norm_layer='instance'),
    'b2m_comb_patch_l1_ctx': dict(no_comb=False, which_gan='patch', objReconLoss='l1', cond_in='ctx')}


def golden_box2mask_variants(only=None, steps=4):
    from oracle import ref_mask_cpu
    base = dict(ndf=16, label_nc=35, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999)
    for tag, extra in BOX2MASK_VARIANTS.items():
        if only and tag not in only:
            continue
        fl = dict(base, **extra)
        ref = ref_shim.box2mask_trainer(**fl)
        ora = ref_mask_cpu.TwoStreamAEMask(**fl)
        assert list(ref.netG.state_dict().keys()) == list(ora.netG.state_dict().keys()), tag
        assert list(ref.netD.state_dict().keys()) == list(ora.netD.state_dict().keys()), tag
        sdG = synth.init_state_dict(ora.netG.state_dict(), 21)
        sdD = synth.init_state_dict(ora.netD.state_dict(), 22)
        for m in (ref, ora):
            m.netG.load_state_dict(sdG)
            m.netD.load_state_dict(sdD)
        rl, ol = [], []
        for s in range(steps):
            b = synth.make_box2mask_batch(s, 0, 2, 64, 64, 35)
            with torch.autograd.graph.allow_mutation_on_saved_tensors():
                r, _ = ref.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'],
                                   b['mask_in'], eval_mode=False)
            rl.append([float(x.detach().reshape(-1)[0]) if torch.is_tensor(x) else float(x) for x in r])
            o = ora.step(b)
            ol.append([o[k] for k in ref_mask_cpu.LOSS_NAMES])
        rl, ol = np.array(rl, np.float64), np.array(ol, np.float64)
        rel = np.abs(rl - ol) / np.maximum(np.abs(rl), 1e-12)
        print('%s: %d steps, max rel(oracle vs reference) per step = %s' % (tag, steps, ' '.join('%.1e' % v for v in rel.max(1))))
        assert rel[:2].max() < 2e-6 and rel.max() < 5e-3, (tag, rel)
        np.savez_compressed(os.path.join(HERE, tag + '.npz'), flags=json.dumps(fl), B=2, H=64, W=64,
                            losses=rl.astype(np.float32), loss_names=np.array(ref_mask_cpu.LOSS_NAMES),
                            g_keys=np.array(list(ref.netG.state_dict().keys())),
                            d_keys=np.array(list(ref.netD.state_dict().keys())))


def golden_box2mask_eval():
    """The evaluation-side methods of the REAL reference's TwoStreamAE_mask on the box2mask_traj configuration (64x64,
    batch 2, ndf 16, seeded weights 21 / 22, batch 0), in this order: generate() (eval mode, fresh running statistics),
    reconstruct(eval_mode=False) (training mode: batch statistics, running statistics move), evaluate() (first sample,
    eval mode on the moved statistics), then ONE training forward() for the [comb_recon_label, obj_recon_label] it returns.
    Per label map: the int64 map, the top-1 value of the blended map and its margin over the runner-up (a comparison may
    skip pixels whose margin is inside fp32 noise)."""
    fl = dict(ndf=16, label_nc=35, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999)
    ref = ref_shim.box2mask_trainer(**fl)
    ref.netG.load_state_dict(synth.init_state_dict(ref.netG.state_dict(), 21))
    ref.netD.load_state_dict(synth.init_state_dict(ref.netD.state_dict(), 22))
    b = synth.make_box2mask_batch(0, 0, 2, 64, 64, 35)
    d = {'label_map': b['label'], 'mask_obj_in': None, 'mask_ctx_in': b['mask_ctx_in'], 'mask_obj_out': None,
         'mask_out': b['mask_out'], 'mask_obj_inst': b['mask_obj_inst'], 'cls': b['cls'], 'mask_in': b['mask_in']}
    out = {}

    def margins(tag, prob_map, gt_mask, gt_one_hot):
        blended = ref.postprocess_output(prob_map.detach(), gt_mask, gt_one_hot)
        top = blended.topk(2, dim=1).values
        out[tag + '_top1'] = top[:, :1].numpy()
        out[tag + '_margin'] = (top[:, :1] - top[:, 1:2]).numpy()

    with torch.no_grad():
        gt_one_hot, input_ctx, gt_mask, cls_onehot, obj_cond = ref.encode_input(b['label'], b['mask_ctx_in'], b['mask_out'],
                                                                            b['mask_in'], b['cls'])
        out.update(enc_onehot_label_sum=gt_one_hot.sum((2, 3)).numpy(), enc_ctx_sum=input_ctx.sum((2, 3)).numpy(),
                   enc_obj_cond_sum=obj_cond.sum((2, 3)).numpy(), enc_cls_argmax=cls_onehot.argmax(1).numpy())
        cond = ref.construct_input_cond(obj_cond, input_ctx)
        ref.netG.set_mode(eval_mode=True)
        _, p_eval, _, _ = ref.netG.forward(cond, cls_onehot)
        ref.netG.set_mode(eval_mode=False)
        margins('generate', p_eval, gt_mask, gt_one_hot)
        g = ref.generate(d)
        assert g['comb_pred_label'].dtype == torch.int64
        out.update(generate_comb=g['comb_pred_label'].numpy(), generate_obj=g['obj_pred_label'].numpy())
        r = ref.reconstruct(d, eval_mode=False)
        margins('reconstruct', r['comb_recon_prob'], gt_mask, gt_one_hot)
        out.update(reconstruct_comb=r['comb_recon_label'].numpy(), reconstruct_obj=r['obj_recon_label'].numpy(),
                   reconstruct_keys=np.array(sorted(r.keys())),
                   running_mean_after=ref.netG.state_dict()['conv_encoder_modules.1.running_mean'].clone().numpy())   # (a copy:
                   # .numpy() aliases the live buffer, which the training step below moves again)
        e = ref.evaluate(d)
        out.update(evaluate_label=e.numpy(), evaluate_cls=b['cls'].numpy())
    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        losses, recon = ref.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'],
                                    b['mask_in'], eval_mode=False)
    out.update(forward_comb=recon[0].detach().numpy(), forward_obj=recon[1].detach().numpy(),
               forward_losses=np.array([float(x.detach().reshape(-1)[0]) if torch.is_tensor(x) else float(x) for x in losses]))
    np.savez_compressed(os.path.join(HERE, 'box2mask_eval.npz'), flags=json.dumps(fl), **out)
    print('box2mask_eval: generate / reconstruct / evaluate / forward outputs of the reference stored; min margins %s' % (
        {k: float(v.min()) for k, v in out.items() if k.endswith('_margin')}))


def golden_data_ops():
    """get_masked_image of the REAL reference (data/base_dataset.py:342-357), sample by sample, for boxes that are
    interior, clipped by the border, empty (hmax == hmin) and full-image; cls2fill 0 and 34."""
    ref_shim.install()
    import importlib
    bd = importlib.import_module('data.base_dataset')
    g = torch.Generator().manual_seed(9)
    image = torch.rand(6, 3, 12, 20, generator=g) * 2 - 1
    bbox = torch.tensor([[3, 2, 11, 9], [0, 0, 20, 12], [5, 4, 5, 9], [15, 7, 20, 12], [0, 3, 7, 3], [1, 1, 2, 2]],
                        dtype=torch.float32)
    out = {'image': image.numpy(), 'bbox': bbox.numpy()}
    for fill in (0, 34):
        ms, os_, cs = [], [], []
        for b in range(image.shape[0]):
            m, o, c = bd.get_masked_image(image[b], bbox[b], cls2fill=fill)
            ms.append(m)
            os_.append(o)
            cs.append(c)
        out['mask_%d' % fill] = torch.stack(ms).numpy()
        out['obj_%d' % fill] = torch.stack(os_).numpy()
        out['ctx_%d' % fill] = torch.stack(cs).numpy()
    np.savez_compressed(os.path.join(HERE, 'data_ops.npz'), **out)
    print('data_ops: get_masked_image pinned on %d boxes' % bbox.shape[0])


ADE = dict(label_nc=49, output_nc=49, norm_layer='instance', add_dilated_layers=True)   # scripts/train_box2mask_ade.sh


def golden_box2mask_ade():
    """The ADE recipe of box2mask (scripts/train_box2mask_ade.sh): label_nc 49, InstanceNorm generator and discriminator,
    two DilatedResnetBlocks (dilation 2 and 4) in front of the latent ResnetBlocks, --lr_control.
    (1) generator forward + parameter gradients of the REAL reference class at 64x64, batch 2 (8x8 latent planes: the
        dilation-4 block works on 2x2 phase images), pinned against the oracle restatement;
    (2) 6 training steps of the REAL trainer with --lr_control (its ``.data[0]`` reads are patched to
        ``.data.reshape(-1)[0]`` in memory: torch >= 0.4 losses are 0-dim), next to the oracle; the (g_lr, d_lr) decisions
        of every step are stored."""
    from oracle import ref_mask_cpu
    ref = ref_shim.box2mask_generator(**ADE)
    ora = ref_mask_cpu.MaskTwoStreamConvSwitchNet(49, 49, norm_layer='instance', add_dilated_layers=True)
    assert list(ref.state_dict().keys()) == list(ora.state_dict().keys()), (
        set(ref.state_dict().keys()) ^ set(ora.state_dict().keys()))
    sd = synth.init_state_dict(ora.state_dict(), 31)
    x = torch.randn(2, 98, 64, 64, generator=torch.Generator().manual_seed(3))
    gy = [torch.randn(2, 49, 64, 64, generator=torch.Generator().manual_seed(5)),
          torch.randn(2, 1, 64, 64, generator=torch.Generator().manual_seed(6))]
    out = dict(x_sum=np.array([x.double().sum().item(), x.double().abs().sum().item()]),
               gy_sum=np.array([gy[0].double().sum().item(), gy[1].double().sum().item()]))
    ref.load_state_dict(sd)
    ora.load_state_dict(sd)
    ref.train()
    ora.train()
    with torch.autograd.graph.allow_mutation_on_saved_tensors():
        a = ref(x.clone(), None)
        ((a[1] * gy[0]).sum() + (a[3] * gy[1]).sum()).backward()
    b = ora(x.clone())
    ((b[1] * gy[0]).sum() + (b[3] * gy[1]).sum()).backward()
    for p, q in zip(a, b):
        assert torch.equal(p, q), (p - q).abs().max()
    out['ctx_prob'] = a[1].detach().numpy()
    out['obj_prob'] = a[3].detach().numpy()
    names, sums = [], []
    gr = dict(ref.named_parameters())
    for k, po in ora.named_parameters():
        gr_k, go_k = gr[k].grad, po.grad
        scale = float(gr_k.abs().max())
        if k.endswith('bias'):
            scale = max(scale, float(gr[k[:-4] + 'weight'].grad.abs().max()))
        assert float((gr_k - go_k).abs().max()) <= 4e-6 * max(scale, 1e-3), (k, scale, float((gr_k - go_k).abs().max()))
        names.append(k)
        sums.append([gr_k.double().sum().item(), gr_k.double().abs().sum().item()])
    out['grad_names'] = np.array(names)
    out['grad_sums'] = np.array(sums)
    np.savez_compressed(os.path.join(HERE, 'box2mask_ade_net.npz'), **out)
    print('box2mask_ade_net: forward bit-exact, gradients pinned (%d parameters)' % len(names))

    fl = dict(ADE, ndf=16, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999, lr_control=True)
    reft = ref_shim.box2mask_trainer(**fl)
    orat = ref_mask_cpu.TwoStreamAEMask(label_nc=49, ndf=16, num_layers_D=3, gan_weight=0.1, lr=0.0002, beta1=0.5, beta2=0.999,
                                        norm_layer='instance', add_dilated_layers=True, lr_control=True)
    assert list(reft.netD.state_dict().keys()) == list(orat.netD.state_dict().keys())
    sdG = synth.init_state_dict(orat.netG.state_dict(), 31)
    sdD = synth.init_state_dict(orat.netD.state_dict(), 32)
    for m in (reft, orat):
        m.netG.load_state_dict(sdG)
        m.netD.load_state_dict(sdD)
    rl, ol, said = [], [], []
    import io
    for s in range(6):
        bt = synth.make_box2mask_batch(s, 0, 2, 64, 64, 49)
        so, sys.stdout = sys.stdout, io.StringIO()
        try:
            with torch.autograd.graph.allow_mutation_on_saved_tensors():
                r, _ = reft.forward(bt['label'], None, bt['mask_ctx_in'], None, bt['mask_out'], bt['mask_obj_inst'],
                                    bt['cls'], bt['mask_in'], eval_mode=False)
            said.append(sys.stdout.getvalue().split('\t')[0].strip())
        finally:
            sys.stdout = so
        rl.append([float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else float(v) for v in r])
        o = orat.step(bt)
        ol.append([o[k] for k in ref_mask_cpu.LOSS_NAMES])
    rl, ol = np.array(rl, np.float64), np.array(ol, np.float64)
    rel = np.abs(rl - ol) / np.maximum(np.abs(rl), 1e-12)
    print('box2mask_ade_traj: lr_control said %s; max rel(oracle vs reference) per step = %s' % (
        said, ' '.join('%.1e' % v for v in rel.max(1))))
    assert rel[:3].max() < 2e-6 and rel.max() < 5e-3, rel
    np.savez_compressed(os.path.join(HERE, 'box2mask_ade_traj.npz'), flags=json.dumps(fl), B=2, H=64, W=64,
                        losses=rl.astype(np.float32), loss_names=np.array(ref_mask_cpu.LOSS_NAMES),
                        lr_control_said=np.array(said))


TINY = dict(model='pix2pixHD_condImg', netG='global', ngf=8, ndf=8, n_downsample_global=4, n_blocks_global=2,
            num_D=2, n_layers_D=3, label_nc=35, no_instance=True)
TINY_GATE = dict(TINY, use_output_gate=True, num_D=3)
TINY_INST = dict(TINY, no_instance=False, n_downsample_global=3)
TINY_TWO = dict(model='pix2pixHD_condImg', netG='global_twostream', ngf=8, ndf=8, n_downsample_global=4,
                n_blocks_global=2, num_D=2, n_layers_D=3, label_nc=35, no_instance=True, no_imgCond=True,
                which_encoder='ctx_label', use_skip=True, use_output_gate=True, mask_gan_input=True)
TINY_COLOR = dict(TINY_TWO, model='pix2pixHD_condImgColor', label_nc=49)
# loss / input flags of options/mask2image_train_options.py:39-46 on a 2-down toy net (the flag sets of
# tests/test_model_gpu.py::test_loss_flag_variants_teacher_forced); --no_ganFeat_loss also changes the discriminator's
# checkpoint keys (getIntermFeat=False: one flattened Sequential per scale, models/Discriminator_NET.py:27-28)
TINY2 = dict(TINY, n_downsample_global=2)
FLAG_VARIANTS = {'tiny_flag_lambda_rec': dict(TINY2, lambda_rec=5.0),
                 'tiny_flag_soft_mask': dict(TINY2, use_soft_mask=True, mask_gan_input=True),
                 'tiny_flag_rec_no_ganfeat': dict(TINY2, lambda_rec=2.0, no_ganFeat_loss=True),
                 'tiny_flag_no_vgg_no_imgcond': dict(TINY2, no_vgg_loss=True, no_imgCond=True),
                 # round 6: the vanilla-GAN loss (nn.BCELoss on Sigmoid outputs, models/losses.py:17-20) -- runnable in the
                 # reference only with --no_ganFeat_loss (with feature matching it drops the Sigmoid, Discriminator_NET.py:24-27)
                 'tiny_flag_no_lsgan': dict(TINY2, no_lsgan=True, no_ganFeat_loss=True)}
# encoder choices of the two-stream generator (--which_encoder ctx | label | ctx_label, models/Pix2Pix_NET.py:126-136;
# 'ctx' is the parser's default and feeds the discriminator the IMAGE ONLY, pix2pixHD_condImg_model.py:70-71,179-180),
# with and without --use_skip / --use_output_gate; 'label' + --use_skip fails inside the reference's own decoder
TINY_TWO3 = dict(model='pix2pixHD_condImg', netG='global_twostream', ngf=8, ndf=8, n_downsample_global=3, n_blocks_global=2,
                 num_D=2, n_layers_D=3, label_nc=35, no_instance=True)
FLAG_VARIANTS.update({
    'tiny_two_ctx': dict(TINY_TWO3, which_encoder='ctx'),
    'tiny_two_ctx_gate_skip': dict(TINY_TWO3, which_encoder='ctx', use_skip=True, use_output_gate=True, no_imgCond=True),
    'tiny_two_ctxlabel_plain': dict(TINY_TWO3, which_encoder='ctx_label'),
    'tiny_two_label': dict(TINY_TWO3, which_encoder='label'),
    'tiny_two_label_gate': dict(TINY_TWO3, which_encoder='label', use_output_gate=True, mask_gan_input=True)})
# round 6 (second half): the generator flags the HIP path used to refuse -- --norm batch (models/layer_util.py:20-21:
# BatchNorm2d(affine=True) in the generator AND the discriminator) and --feat_fusion early_concat | late_add | late_concat
# (models/Pix2Pix_NET.py:133-144,212-217; layer_util.py:295-330).  Three blocks: late fusion splits them 1 + 1 | 2.
FLAG_VARIANTS.update({
    'tiny_flag_norm_batch': dict(TINY2, norm='batch'),
    'tiny_two_early_concat': dict(TINY_TWO3, which_encoder='ctx_label', feat_fusion='early_concat'),
    'tiny_two_late_add': dict(TINY_TWO3, which_encoder='ctx_label', feat_fusion='late_add', n_blocks_global=3),
    'tiny_two_late_concat_batch': dict(TINY_TWO3, which_encoder='ctx_label', feat_fusion='late_concat', norm='batch',
                                       n_blocks_global=3, use_skip=True, use_output_gate=True, no_imgCond=True,
                                       mask_gan_input=True)})
C1 = dict(model='pix2pixHD_condImg', netG='global', ngf=64, ndf=64, n_downsample_global=4, n_blocks_global=9,
          num_D=1, n_layers_D=3, label_nc=35, no_instance=True)
C2 = dict(C1, num_D=3)
# BASELINE config 4: ADE20K-shaped 256x256, colour two-stream path (the shipped recipe flags of
# scripts/train_mask2image_ade.sh + the colour model), label_nc 49, num_D 2
C4 = dict(model='pix2pixHD_condImgColor', netG='global_twostream', ngf=64, ndf=64, n_downsample_global=4,
          n_blocks_global=9, num_D=2, n_layers_D=3, label_nc=49, no_instance=True, no_imgCond=True,
          which_encoder='ctx_label', use_skip=True, use_output_gate=True, mask_gan_input=True)

if __name__ == '__main__':
    what = sys.argv[1:] or ['tiny']
    torch.manual_seed(0)
    if 'tiny' in what:
        golden_nets()
        trajectory('tiny_global', TINY, 2, 32, 64, 20, save_outputs=True)
        trajectory('tiny_gate3', TINY_GATE, 2, 32, 64, 5)
        trajectory('tiny_inst', TINY_INST, 2, 32, 64, 5, save_outputs=True)
        trajectory('tiny_twostream', TINY_TWO, 2, 64, 64, 20, save_outputs=True)
        trajectory('tiny_color', TINY_COLOR, 2, 64, 64, 5, color=True)
    if 'flags' in what or any(w in FLAG_VARIANTS for w in what):
        for tag, fl in FLAG_VARIANTS.items():
            if 'flags' not in what and tag not in what:
                continue
            trajectory(tag, fl, 2, 64 if fl['netG'] == 'global_twostream' else 32, 64, 5, save_keys=True)
    if 'c1' in what:
        trajectory('c1_traj', C1, 1, 128, 256, 20)
    if 'c2' in what:
        trajectory('c2_traj', C2, 8, 256, 512, 20)
    if 'box2mask' in what:
        golden_box2mask_net()
        golden_box2mask_traj()
    if 'box2mask_variants' in what or any(w in BOX2MASK_VARIANTS for w in what):
        golden_box2mask_variants(None if 'box2mask_variants' in what else what)
    if 'box2mask_eval' in what:
        golden_box2mask_eval()
    if 'data_ops' in what:
        golden_data_ops()
    if 'box2mask_ade' in what:
        golden_box2mask_ade()
    if 'c4' in what:
        trajectory('c4_traj', C4, 4, 256, 256, 3, color=True)   # bs 4 of the bs-16 config keeps it to minutes
