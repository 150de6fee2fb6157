"""Import harness for the *real* reference (``/root/reference``)  --  TEST INFRASTRUCTURE.

Works only in the build container (the reference does not travel to the GPU box).  Used by
``tests/golden/make_golden.py`` to (1) validate ``oracle/ref_cpu.py`` against the reference's own
classes and (2) emit the golden vectors committed under ``tests/golden/``.  Nothing in
``/root/reference`` is modified: the shim emulates Python-2 implicit relative imports through
``sys.path``, stubs the absent ``torchvision``/``cStringIO`` modules and makes ``.cuda()`` an identity.
"""
import importlib
import io
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get('HIM_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'models'))


_installed = False


def install():
    global _installed
    if _installed:
        return
    assert available(), 'reference checkout not present'
    sys.dont_write_bytecode = True      # importing must not leave __pycache__ directories inside the reference tree
    sys.path[:0] = [REF, os.path.join(REF, 'models'), os.path.join(REF, 'options')]
    cs = types.ModuleType('cStringIO')
    cs.StringIO = io.BytesIO
    sys.modules['cStringIO'] = cs
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvt = types.ModuleType('torchvision.transforms')

    def vgg19(pretrained=False):
        cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M',
               512, 512, 512, 512, 'M']
        layers, c = [], 3
        for v in cfg:
            if v == 'M':
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(True)]
                c = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    tvm.vgg19 = vgg19

    # torchvision.transforms is absent from the image; the reference's loader (data/base_dataset.py:236-268) uses four
    # of its classes.  Restated from torchvision's documented behaviour (0.2 ... 0.20 agree on these): Compose applies
    # in order, Lambda calls, ToTensor = HWC bytes -> CHW float / 255 (integer-mode images keep their integers),
    # Normalize = (t - mean) / std per channel.
    import numpy as _np

    class Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class Lambda(object):
        def __init__(self, lambd):
            self.lambd = lambd

        def __call__(self, x):
            return self.lambd(x)

    class ToTensor(object):
        def __call__(self, pic):
            if pic.mode in ('I', 'I;16'):
                a = _np.array(pic, _np.int32)        # torchvision: int32 / int16; ids below 32768 agree
            elif pic.mode == 'F':
                a = _np.array(pic, _np.float32)
            elif pic.mode == '1':
                a = 255 * _np.array(pic, _np.uint8)
            else:
                a = _np.array(pic, _np.uint8)
            if a.ndim == 2:
                a = a[:, :, None]
            t = torch.from_numpy(_np.ascontiguousarray(a.transpose(2, 0, 1)))
            return t.float().div(255) if t.dtype == torch.uint8 else t

    class Normalize(object):
        def __init__(self, mean, std):
            self.mean, self.std = mean, std

        def __call__(self, t):
            t = t.clone()
            for c in range(t.size(0)):
                t[c].sub_(self.mean[c]).div_(self.std[c])
            return t

    tvt.Compose, tvt.Lambda, tvt.ToTensor, tvt.Normalize = Compose, Lambda, ToTensor, Normalize
    tv.models, tv.transforms = tvm, tvt
    sys.modules.update({'torchvision': tv, 'torchvision.models': tvm, 'torchvision.transforms': tvt})
    torch.Tensor.cuda = lambda s, *a, **k: s
    nn.Module.cuda = lambda s, *a, **k: s
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.ByteTensor = torch.ByteTensor
    torch.cuda.set_device = lambda *a, **k: None
    _installed = True


def nets():
    """(Pix2Pix_NET, Discriminator_NET, losses, sn_utils, layer_util) reference modules."""
    install()
    return tuple(importlib.import_module(m) for m in
                 ('Pix2Pix_NET', 'Discriminator_NET', 'losses', 'sn_utils', 'layer_util'))


def make_opt(argv):
    install()
    from options.mask2image_train_options import MaskToImageTrainOptions
    stdout = sys.stdout
    sys.stdout = io.StringIO()
    try:
        opt = MaskToImageTrainOptions().parse(save=False, default_args=argv)
    finally:
        sys.stdout = stdout
    return opt


def make_model(argv, color=False):
    """Construct the reference's Pix2PixHDModel_condImg[Color] on CPU."""
    install()
    opt = make_opt(argv)
    name = 'models.pix2pixHD_condImgColor_model' if color else 'models.pix2pixHD_condImg_model'
    M = importlib.import_module(name)
    cls = M.Pix2PixHDModel_condImgColor if color else M.Pix2PixHDModel_condImg
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    stdout = sys.stdout
    sys.stdout = io.StringIO()
    try:
        model = cls(opt)
    finally:
        sys.stdout = stdout
        torch.cuda.is_available = lambda: False
    model._restore_cuda_available = real
    return model, opt


def ref_step(model, batch, color=False):
    """train_mask2image.py:58-86 executed on the reference model object; returns the 5 losses."""
    kw = dict(label=batch['label'], inst=batch['inst'], image=batch['image'], feat=None,
              mask_in=batch['mask_in'], mask_out=batch['mask_out'], infer=False)
    if color:
        kw['obj_mask'] = batch['obj_mask']
    losses, _ = model(**kw)
    losses = [torch.mean(x) if not isinstance(x, int) else x for x in losses]
    ld = dict(zip(model.loss_names, losses))
    loss_D = (ld['D_fake'] + ld['D_real']) * 0.5
    loss_G = ld['G_GAN'] + ld['G_GAN_Feat'] + ld['G_VGG']
    model.optimizer_G.zero_grad()
    loss_G.backward()
    model.optimizer_G.step()
    model.optimizer_D.zero_grad()
    loss_D.backward()
    model.optimizer_D.step()
    return {k: float(v.detach()) for k, v in ld.items()}


# ------------------------------------------------------------------------------------------------------------------
# box2mask (second hot path): the generator classes are Python-2 code (xrange, dict.iteritems, int '/')
# ------------------------------------------------------------------------------------------------------------------
def _load_patched(name, subs):
    """Import reference module ``models/<name>.py`` with textual Python-2 -> 3 substitutions applied IN MEMORY (the file
    under /root/reference is not touched and nothing of it is written to disk)."""
    path = os.path.join(REF, 'models', name + '.py')
    with open(path) as f:
        src = f.read()
    for a, b in subs:
        src = src.replace(a, b)
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src, path, 'exec'), mod.__dict__)
    return mod


def box2mask_generator(**flags):
    """The reference's MaskTwoStreamConvSwitch_NET (models/MaskTwoStreamConvSwitch_NET.py) built on CPU from the flags
    of scripts/train_box2mask_city.sh (overridable)."""
    install()
    import builtins
    builtins.xrange = range
    importlib.import_module('layer_util')
    py3 = [('.iteritems()', '.items()'), ('output_dim = input_dim/2', 'output_dim = input_dim//2')]
    _load_patched('MaskContextAE_NET', py3)
    M = _load_patched('MaskTwoStreamConvSwitch_NET', py3)
    d = dict(label_nc=35, output_nc=35, fineSize=256, num_layers=3, conv_dim=64, conv_size=4, embed_dim=1024, z_dim=512,
             norm_layer='batch', use_dropout=False, skip_start=1, skip_end=3, use_resnetblock=1, num_resnetblocks=1,
             fusion_type='add', first_conv_stride=1, first_conv_size=5, which_stream='obj_context', cond_in='ctx_obj',
             use_simpleRes=False, n_blocks=6, add_dilated_layers=False)
    d.update(flags)
    net = M.MaskTwoStreamConvSwitch_NET(types.SimpleNamespace(**d))
    net.initialize()
    return net


def box2mask_trainer(**flags):
    """The reference's TwoStreamAE_mask (models/TwoStreamAE_mask.py) on CPU with the flags of
    scripts/train_box2mask_city.sh (overridable).  Its forward() runs the whole training step (G and D Adam included)."""
    install()
    import builtins
    builtins.xrange = range
    importlib.import_module('layer_util')
    py3 = [('.iteritems()', '.items()'), ('output_dim = input_dim/2', 'output_dim = input_dim//2')]
    _load_patched('MaskContextAE_NET', py3)
    _load_patched('MaskTwoStreamConvSwitch_NET', py3)
    _load_patched('MaskTwoStreamConv_NET', py3)          # the parser's default (no --no_comb)
    bm = _load_patched('base_model', py3)
    sys.modules['models.base_model'] = bm
    if not hasattr(nn, 'NLLLoss2d'):
        nn.NLLLoss2d = nn.NLLLoss
    # lr_control (Discriminator_NET.py:190-211) indexes its losses with .data[0]; torch >= 0.4 losses are 0-dim tensors
    _load_patched('Discriminator_NET', py3 + [('.data[0]', '.data.reshape(-1)[0]')])
    T = _load_patched('TwoStreamAE_mask', py3)
    d = dict(label_nc=35, output_nc=35, fineSize=256, num_layers=3, conv_dim=64, conv_size=4, embed_dim=1024, z_dim=512,
             norm_layer='batch', use_dropout=False, skip_start=1, skip_end=3, use_resnetblock=1, num_resnetblocks=1,
             fusion_type='add', first_conv_stride=1, first_conv_size=5, which_stream='obj_context', cond_in='ctx_obj',
             use_simpleRes=False, n_blocks=6, add_dilated_layers=False, no_comb=True, use_gan=True,
             which_gan='patch_multiscale', gan_weight=0.1, rec_weight=1.0, use_output_gate=True, ndf=64, num_layers_D=3,
             objReconLoss='bce', use_ganFeat_loss=True, lambda_feat=1.0, lr=0.0002, beta1=0.5, beta2=0.999,
             lr_control=False, isTrain=True, gpu_ids=[0], checkpoints_dir='/tmp/him_b2m', name='g', resize_or_crop='none',
             continue_train=False, load_pretrain='', which_epoch='latest', niter=400, niter_decay=0)
    d.update(flags)
    real = torch.cuda.is_available
    torch.cuda.is_available = lambda: True
    stdout = sys.stdout
    sys.stdout = io.StringIO()
    try:
        model = T.TwoStreamAE_mask(types.SimpleNamespace(**d))
    finally:
        sys.stdout = stdout
        torch.cuda.is_available = lambda: False
    return model
