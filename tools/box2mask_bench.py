#!/usr/bin/env python
"""Throughput of the box2mask training step (BASELINE config 5 shape: 256x256, label_nc 35, the flags of
scripts/train_box2mask_city.sh) on one MI355X, plus a steady-state memory check.  Not the bench.py line."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

BS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = create_model(dict(model='AE_maskgen_twostream', gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/b2m', name='b'))
bs = [synth.make_box2mask_batch(i, 0, BS, 256, 256, 35) for i in range(2)]
bs = [{k: (v.cuda() if k != 'cls' else v) for k, v in b.items()} for b in bs]


def step(b):
    return m.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'], b['mask_in'])[0]


for i in range(4):
    step(bs[i % 2])
torch.cuda.synchronize()
a0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
n = 10
for i in range(n):
    ld = step(bs[i % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print('box2mask 256x256 bs %d: %.2f ms/step = %.1f images/s; live memory drift over %d steps: %.1f MB; losses %s' % (
    BS, dt * 1e3, BS / dt, n, (torch.cuda.memory_allocated() - a0) / 2 ** 20,
    [round(float(x.reshape(-1)[0]) if torch.is_tensor(x) else float(x), 4) for x in ld]))
