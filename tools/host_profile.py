"""cProfile of the host side of optimize_parameters (is the step launch-bound?)."""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

m = create_model(dict(bench.C2, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/x', name='b', batchSize=8))
b = {k: v.cuda() for k, v in synth.make_batch(0, 0, 8, 256, 512).items()}
for i in range(4):
    m.optimize_parameters(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    m.optimize_parameters(b)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue %.1f ms/step, wall %.1f ms/step' % ((t1 - t0) / 5 * 1e3, (t2 - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(3):
    m.optimize_parameters(b)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
