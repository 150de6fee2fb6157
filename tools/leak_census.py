"""Which CUDA tensors accumulate from one training step to the next?"""
import collections
import gc
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

m = create_model(dict(bench.C2, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/x', name='b', batchSize=8))
b = {k: v.cuda() for k, v in synth.make_batch(0, 0, 8, 256, 512).items()}


def census():
    c = collections.Counter()
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                c[(tuple(o.shape), o.grad_fn.__class__.__name__ if o.grad_fn is not None else '-')] += 1
        except Exception:
            pass
    return c


for i in range(3):
    m.optimize_parameters(b)
torch.cuda.synchronize()
c0 = census()
a0 = torch.cuda.memory_allocated()
for i in range(2):
    m.optimize_parameters(b)
torch.cuda.synchronize()
c1 = census()
print('allocated grew %.2f GB over 2 steps' % ((torch.cuda.memory_allocated() - a0) / 1e9))
diff = {k: c1[k] - c0.get(k, 0) for k in c1 if c1[k] != c0.get(k, 0)}
for k, v in sorted(diff.items(), key=lambda kv: -abs(kv[1]))[:40]:
    print(v, k)
