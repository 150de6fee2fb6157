"""Build-owned deterministic data and weights (counter-based Philox, never torch's global RNG).

The same arrays are produced in the build container, on the GPU box and in the oracle, so a 730 MB
generator never has to travel: parity tests load *these* tensors into both sides.

* ``make_batch``  -- Cityscapes-shaped synthetic batches with the dict keys ``SegmentationDataset``
  emits (reference ``data/segmentation_dataset.py:121-130``): label ids as float, zeros ``inst``,
  image in [-1,1), box masks.
* ``init_state_dict`` -- the reference's *effective* initial state (``models/layer_util.py:9-16``:
  conv weights ~ N(0, 0.02); biases keep torch's default U(+-1/sqrt(fan_in))).
"""
from collections import OrderedDict

import numpy as np
import torch


def _gen(seed, idx=0):
    return np.random.Generator(np.random.Philox(key=np.array([seed, idx], dtype=np.uint64)))


def make_batch(step, rank, B, H, W, label_nc=35, color=False, block=16):
    """Seed = 1000 + step*64 + rank."""
    g = _gen(1000 + step * 64 + rank)
    gh, gw = max(H // block, 1), max(W // block, 1)
    ids = g.integers(0, max(label_nc, 1), size=(B, 1, gh, gw)).astype(np.float32)
    label = np.repeat(np.repeat(ids, H // gh, axis=2), W // gw, axis=3)
    image = (g.random((B, 3, H, W), dtype=np.float32) * 2.0 - 1.0).astype(np.float32)
    mask_in = np.zeros((B, 1, H, W), np.float32)
    mask_in[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4] = 1.0
    hh, hw = 0.5 * 1.35 * (H / 2.0), 0.5 * 1.35 * (W / 2.0)
    y0, y1 = max(int(round(H / 2.0 - hh)), 0), min(int(round(H / 2.0 + hh)), H)
    x0, x1 = max(int(round(W / 2.0 - hw)), 0), min(int(round(W / 2.0 + hw)), W)
    mask_out = np.zeros((B, 1, H, W), np.float32)
    mask_out[:, :, y0:y1, x0:x1] = 1.0
    out = OrderedDict(label=label, inst=np.zeros((B, 1, H, W), np.float32), image=image,
                      mask_in=mask_in, mask_out=mask_out)
    if color:
        out['obj_mask'] = mask_in.copy()
    return OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in out.items())


def make_box2mask_batch(step, rank, B, H, W, label_nc=35):
    """Synthetic box2mask sample (keys of data/segmentation_dataset.py for the box2mask loader): piecewise-constant label
    map, a centred object box ``mask_in``, the dilated context box ``mask_out``, the context map (label outside
    ``mask_out``, the last class id inside), the object's class = the label at the centre, and its instance mask inside
    the box."""
    base = make_batch(step, rank, B, H, W, label_nc)
    label, mask_in, mask_out = base['label'], base['mask_in'], base['mask_out']
    cls = label[:, :, H // 2, W // 2].long().view(B, 1)
    ctx = label * (1.0 - mask_out) + float(label_nc - 1) * mask_out
    inst = (label == cls.view(B, 1, 1, 1).float()).float() * mask_in
    return OrderedDict(label=label, mask_in=mask_in, mask_out=mask_out, mask_ctx_in=ctx, cls=cls, mask_obj_inst=inst)


def init_state_dict(shapes, seed, kind='gan'):
    """``shapes``: ordered name -> shape (a module's ``state_dict()`` works).  kind 'gan': weights
    N(0,0.02), bias U(+-1/sqrt(fan_in)); kind 'vgg': He-normal weights, zero bias (synthetic VGG)."""
    out = OrderedDict()
    names = list(shapes.keys())
    for idx, name in enumerate(names):
        shape = tuple(shapes[name].shape) if hasattr(shapes[name], 'shape') else tuple(shapes[name])
        g = _gen(seed, idx)
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith('weight') and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            std = 0.02 if kind == 'gan' else float(np.sqrt(2.0 / fan_in))
            arr = g.standard_normal(n, dtype=np.float32) * np.float32(std)
        elif name.endswith('num_batches_tracked'):          # BatchNorm2d bookkeeping (int64 scalar)
            out[name] = torch.zeros(shape, dtype=torch.int64)
            continue
        elif name.endswith('running_var'):                  # BatchNorm2d statistics: positive, around 1
            arr = np.float32(1.0) + np.abs(g.standard_normal(n, dtype=np.float32)) * np.float32(0.1)
        elif name.endswith('running_mean'):
            arr = g.standard_normal(n, dtype=np.float32) * np.float32(0.1)
        elif name.endswith('weight') and len(shape) == 1:   # BatchNorm2d gamma ~ N(1, 0.02) (layer_util.py:14-15)
            arr = np.float32(1.0) + g.standard_normal(n, dtype=np.float32) * np.float32(0.02)
        elif name.endswith('bias') and len(tuple(shapes[name[:-4] + 'weight'].shape)
                                           if hasattr(shapes[name[:-4] + 'weight'], 'shape')
                                           else tuple(shapes[name[:-4] + 'weight'])) == 1:
            arr = (g.random(n, dtype=np.float32) * 2.0 - 1.0) * np.float32(0.1)   # BatchNorm2d beta (non-zero on purpose)
        elif name.endswith('bias'):
            wname = name[:-4] + 'weight'
            ws = tuple(shapes[wname].shape) if hasattr(shapes[wname], 'shape') else tuple(shapes[wname])
            bound = 1.0 / np.sqrt(ws[1] * ws[2] * ws[3])
            if kind == 'vgg':
                arr = np.zeros(n, np.float32)
            else:
                arr = ((g.random(n, dtype=np.float32) * 2.0 - 1.0) * np.float32(bound)).astype(np.float32)
        elif name.endswith('.u') or name == 'u':
            arr = g.standard_normal(n, dtype=np.float32)
        else:
            arr = g.standard_normal(n, dtype=np.float32)
        out[name] = torch.from_numpy(arr.reshape(shape).astype(np.float32))
    return out
