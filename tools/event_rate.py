"""Per-step largest gradient distance from float64 of the HIP path, the host fp32 oracle and torch's GPU fp32 operators over N
teacher-forced steps of a toy golden configuration (profiles/r06_ab_log.txt section 6):
    python tools/event_rate.py tiny_two_early_concat 24      (on the GPU box)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_model_gpu as T
tag, steps = sys.argv[1], int(sys.argv[2])
try:
    T._teacher_forced(tag, steps, k_typical=T.PARITY_K_TYPICAL, out_tag='evrate_' + tag)
except AssertionError as e:
    print('ASSERT', str(e)[:300])
r = json.load(open(os.path.join(T.OUT, 'teacher_forced_evrate_%s.json' % tag)))
d = r['grad_distance_from_fp64']; names = d['tensors']
for net in 'GD':
    idx = [i for i, n in enumerate(names) if n.startswith(net)]
    for who in ('hip', 'oracle_live', 'torch_gpu_fp32'):
        a = np.array(d[who])[:, idx]
        print(net, who, 'per-step max:', ' '.join('%.0e' % v for v in a.max(1)))
