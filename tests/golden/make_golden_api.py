"""tests/golden/api_surface.json: the Python surface of the drop-in boundary read from the REAL reference with
``inspect`` -- for every class a trainer script or a user's own model code touches: method names and their positional
parameter names in order.  Names only (data, no source).  Build container only (oracle/ref_shim.py).

    python tests/golden/make_golden_api.py
"""
import importlib
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from oracle import ref_shim                                              # noqa: E402

# reference module:class -> module (under the build's package) holding the class of the same name
CLASSES = {
    'models.base_model:BaseModel': 'models.base_model',
    'models.pix2pixHD_condImg_model:Pix2PixHDModel_condImg': 'models.pix2pixHD_condImg_model',
    'models.pix2pixHD_condImgColor_model:Pix2PixHDModel_condImgColor': 'models.pix2pixHD_condImgColor_model',
    'Pix2Pix_NET:GlobalGenerator': 'models.Pix2Pix_NET',
    'Pix2Pix_NET:LocalEnhancer': 'models.Pix2Pix_NET',
    'Pix2Pix_NET:GlobalTwoStreamGenerator': 'models.Pix2Pix_NET',
    'Discriminator_NET:MultiscaleDiscriminator': 'models.Discriminator_NET',
    'layer_util:ResnetBlock': 'models.layer_util',
    'layer_util:Vgg19': 'models.layer_util',
    'losses:GANLoss': 'models.losses',
    'losses:VGGLoss': 'models.losses',
    'sn_utils:SNConv2d': 'models.sn_utils',
    'sn_utils:SNLinear': 'models.sn_utils',
}


def surface(cls):
    """{method: [positional parameter names]} of the methods the class's own module family defines (not torch's)."""
    out = {}
    for name, fn in inspect.getmembers(cls, inspect.isfunction):
        if name.startswith('__') and name not in ('__init__', '__call__'):
            continue
        if (fn.__module__ or '').split('.')[0] in ('torch', 'oracle'):     # torch.nn.Module's own / the shim's .cuda()
            continue
        out[name] = list(inspect.signature(fn).parameters)
    return out


if __name__ == '__main__':
    ref_shim.install()
    res = {}
    for ref, mine in CLASSES.items():
        mod, cls = ref.split(':')
        res[cls] = dict(build_module=mine, methods=surface(getattr(importlib.import_module(mod), cls)))
    # box2mask trainer: Python-2 source, imported through the shim's in-memory substitutions
    trainer = type(ref_shim.box2mask_trainer())
    res[trainer.__name__] = dict(build_module='models.TwoStreamAE_mask', methods=surface(trainer))
    # truth table of the host-side lr_control gate (models/Discriminator_NET.py:190-211) from the reference's own function
    import io
    import torch
    D = importlib.import_module('Discriminator_NET')
    grid = [0.0, 0.1, 0.29, 0.3, 0.31, 0.5, 0.69, 0.7, 0.71, 0.9, 1.2]
    table = []
    stdout, sys.stdout = sys.stdout, io.StringIO()
    try:
        for r in grid:
            for fk in grid:
                g_lr, d_lr = D.lr_control(torch.tensor([0.5]), torch.tensor([r]), torch.tensor([fk]))
                table.append([r, fk, g_lr, d_lr])
    finally:
        sys.stdout = stdout
    with open(os.path.join(HERE, 'lr_control_table.json'), 'w') as f:
        json.dump(dict(note='[loss_D_real, loss_D_fake, g_lr, d_lr] of the reference lr_control, gan_margin 0.3', rows=table), f)
    with open(os.path.join(HERE, 'api_surface.json'), 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print({k: len(v['methods']) for k, v in res.items()})
