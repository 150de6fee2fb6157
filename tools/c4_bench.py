#!/usr/bin/env python
"""Throughput of BASELINE config 4 (ADE20K-shaped 256x256, colour two-stream generator, label_nc 49, num_D 2, bs 16) --
not the bench.py line (that is config C2), reported in DESIGN.md for the second generator family."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

C4 = dict(model='pix2pixHD_condImgColor', netG='global_twostream', ngf=64, ndf=64, n_downsample_global=4,
          n_blocks_global=9, num_D=2, n_layers_D=3, label_nc=49, no_instance=True, no_imgCond=True,
          which_encoder='ctx_label', use_skip=True, use_output_gate=True, mask_gan_input=True)
BS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = create_model(dict(C4, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/c4', name='b', batchSize=BS))
bs = [{k: v.cuda() for k, v in synth.make_batch(i, 0, BS, 256, 256, 49, True).items()} for i in range(2)]
for i in range(4):
    m.optimize_parameters(bs[i % 2])
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for i in range(n):
    ld = m.optimize_parameters(bs[i % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('C4 two-stream colour 256x256 bs %d: %.2f ms/step = %.1f images/s   losses %s' % (
    BS, dt * 1e3, BS / dt, {k: round(float(v.detach()), 4) for k, v in ld.items()}))
