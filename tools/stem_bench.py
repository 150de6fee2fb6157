#!/usr/bin/env python
"""Stem conv7x7 (38 -> 64 @256x512, bs 8): dense MFMA path vs the one-hot (label-id) path, fwd and wgrad."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import ops, synth
from conv_bench import timeit

b = {k: v.cuda() for k, v in synth.make_batch(0, 0, 8, 256, 512).items()}
buf, n_label, n_cond = ops.encode_channels(b['label'], b['inst'], b['image'], b['mask_in'], 35, False)
w = (torch.randn(64, 38, 7, 7, device='cuda') * 0.02).requires_grad_(True)
bias = torch.zeros(64, device='cuda', requires_grad=True)
for tag, x in (('one-hot', buf), ('dense', buf.clone())):
    y = ops.conv2d(x, w, bias, 1, 3, 'reflect', 'none')
    gy = torch.randn_like(y)
    with torch.no_grad():
        tf = timeit(lambda: ops.conv2d(x, w, bias, 1, 3, 'reflect', 'none'))
    tw = timeit(lambda: torch.autograd.grad(y, (w, bias), gy, retain_graph=True))
    print('%-8s %s  fwd %.3f ms  wgrad %.3f ms' % (tag, y.grad_fn.__class__.__name__, tf, tw))
