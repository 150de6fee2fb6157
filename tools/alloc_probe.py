"""Does the caching allocator reach a steady state (no hipMalloc/hipFree per step)?"""
import os
import sys
import time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.models import create_model

m = create_model(dict(bench.C2, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/x', name='b', batchSize=8))
bs = [{k: v.cuda() for k, v in synth.make_batch(i, 0, 8, 256, 512).items()} for i in range(2)]
import gc
keys = ['allocated_bytes.all.current', 'num_device_alloc', 'num_device_free', 'num_alloc_retries', 'reserved_bytes.all.current',
        'allocated_bytes.all.peak', 'num_sync_all_streams']
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.optimize_parameters(bs[i % 2])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if os.environ.get('PROBE_GC') and i % 3 == 2:
        print('gc.collect ->', gc.collect())
    st = torch.cuda.memory_stats()
    print('step %d host %.1f ms wall %.1f ms  ' % (i, (t1 - t0) * 1e3, (t2 - t0) * 1e3) +
          '  '.join('%s=%s' % (k.split('.')[0], st.get(k)) for k in keys))
