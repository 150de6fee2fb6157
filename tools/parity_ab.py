"""A/B of the toy teacher-forced parity run under different kernel selections (python tools/parity_ab.py [tag] [steps])."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402

torch.set_num_threads(min(32, torch.get_num_threads()))
import test_model_gpu as T  # noqa: E402
from neurips18_hierchical_image_manipulation_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'tiny_twostream'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
VARIANTS = [('pinned', {}), ('no F(4x4)', dict(wino4_min_c=-1)), ('no fused Winograd', dict(wino_fused_min_c=-1)),
            ('no F(4x4), no fused', dict(wino4_min_c=-1, wino_fused_min_c=-1)), ('direct form', dict(wino_min_c=-1))]
for name, over in VARIANTS:
    algo = dict(T.PINNED_ALGO)
    algo.update(over)
    with ops.algo_scope(**algo):
        try:
            T._teacher_forced_run(tag, steps, T.PARITY_LOSS_TOL, None, None, None, None, 2.0, 'ab', steps, 0)
            res = 'pass'
        except AssertionError as e:
            res = 'FAIL ' + str(e)[:160]
    rep = json.load(open(os.path.join(T.OUT, 'teacher_forced_ab.json')))
    print('%-22s %s' % (name, res))
    print('      G grad median per step  hip:', ' '.join('%.0e' % m['G_grad_hip'] for m in rep['median_over_tensors_per_step']))
    print('                           oracle:', ' '.join('%.0e' % m['G_grad_oracle'] for m in rep['median_over_tensors_per_step']))
    sys.stdout.flush()
