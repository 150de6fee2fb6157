"""Where does a teacher-forced parity step at C2 spend its wall time?  (python tools/parity_time_probe.py [cpu])"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
if len(sys.argv) > 1 and sys.argv[1] == 'cpu':
    os.environ['HIM_FP64_ORACLE_CPU'] = '1'
import torch  # noqa: E402

torch.set_num_threads(min(32, torch.get_num_threads()))
import fp64_anchor as fa  # noqa: E402
import test_model_gpu as T  # noqa: E402
from neurips18_hierchical_image_manipulation_amd import synth  # noqa: E402

g = T.load_golden('c2_traj')
flags = json.loads(str(g['flags']))
B, H, W = int(g['B']), int(g['H']), int(g['W'])
t = [time.perf_counter()]


def lap(name):
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    print('%-34s %7.2f s' % (name, t[-1] - t[-2]), flush=True)


model, om = T.build(flags), fa.make_oracle(flags)
lap('build HIP model + fp32 oracle')
om64 = fa.make_oracle(flags, torch.float64)
lap('build fp64 oracle (%s)' % fa.fp64_device())
for s in range(2):
    T._adopt(model, om)
    lap('adopt (HIP <- oracle)')
    fa.adopt64(om64, om)
    lap('adopt64')
    before = fa.snapshot(om)
    lap('snapshot')
    b = synth.make_batch(s, 0, B, H, W, flags.get('label_nc', 35), False)
    got = model.optimize_parameters(b)
    model.sync()
    lap('HIP step')
    ref = om.optimize_parameters(b)
    lap('fp32 oracle step (CPU)')
    q_hip, q32 = T._hip_quantities(model, before), fa.oracle_quantities(om, before)
    lap('quantities')
    fa.step64(om64, b)
    lap('fp64 step (%s)' % fa.fp64_device())
    q64 = fa.oracle_quantities(om64, before)
    names = list(q32)
    e1 = {n: {'grad': fa.rel_l2(q_hip[n]['grad'], q64[n]['grad']), 'delta': fa.rel_l2(q_hip[n]['delta'], q64[n]['delta'])} for n in names}
    lap('rel_l2 HIP vs fp64')
    e2 = {n: {'grad': fa.rel_l2(q32[n]['grad'], q64[n]['grad']), 'delta': fa.rel_l2(q32[n]['delta'], q64[n]['delta'])} for n in names}
    lap('rel_l2 oracle vs fp64')
