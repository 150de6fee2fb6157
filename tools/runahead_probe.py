"""How far does the host run ahead of the GPU?  Per step: the host time at which optimize_parameters() returned and the GPU
time at which the main stream reached that point (event), both relative to the start of the loop."""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

m = create_model(dict(bench.C2, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/x', name='b', batchSize=8))
bs = [{k: v.cuda() for k, v in synth.make_batch(s, 0, 8, 256, 512).items()} for s in range(4)]
torch.cuda.synchronize()
ready = torch.cuda.Event()
ready.record()
for b in bs:
    b['ready_event'] = ready
for i in range(5):
    m.optimize_parameters(bs[i % 4])
torch.cuda.synchronize()
N = 12
base = torch.cuda.Event(enable_timing=True)
base.record()
t0 = time.perf_counter()
evs, hosts, hosts_fwd = [], [], []
orig_forward = m.forward


def fwd(*a, **k):
    out = orig_forward(*a, **k)
    hosts_fwd.append(time.perf_counter() - t0)
    return out


m.forward = fwd
for i in range(N):
    m.optimize_parameters(bs[i % 4])
    hosts.append(time.perf_counter() - t0)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    evs.append(e)
torch.cuda.synchronize()
print('step   host: forward enqueued / step enqueued (ms)   GPU main stream reached the end of the step (ms)')
for i in range(N):
    print('%4d   %8.1f %8.1f   %8.1f' % (i, hosts_fwd[i] * 1e3, hosts[i] * 1e3, base.elapsed_time(evs[i])))
