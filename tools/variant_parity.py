#!/usr/bin/env python
"""Parity REPORT (not a test) of an opt-in reduced-work variant: the C2 teacher-forced run of
tests/test_model_gpu.py -- 6 steps, float64 anchor, per-tensor ratios -- with the variant's HimAlgo bit set.  The bounds of the
shipped build are evaluated and RECORDED, not asserted: the variant is allowed to exceed the per-tensor protocol (that is why
it is not the headline), north_star's own bar (losses to 1e-3 per step) is what it has to hold.

    python tools/variant_parity.py f4x4-resblock-fwd        -> gpurun_out/teacher_forced_c2_traj_<variant>.json + a summary
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    variant = sys.argv[1] if len(sys.argv) > 1 else 'f4x4-resblock-fwd'
    import test_model_gpu as T
    import fp64_anchor as fa  # noqa: F401
    from neurips18_hierchical_image_manipulation_amd import ops
    from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_WINO4_TRAIN_FWD
    bits = {'f4x4-resblock-fwd': ALGO_WINO4_TRAIN_FWD}[variant]
    algo = dict(T.PINNED_ALGO, disable=bits)
    tag = 'c2_traj_' + variant.replace('-', '_')
    failed = None
    with ops.algo_scope(**algo):
        try:
            T._teacher_forced_run('c2_traj', 6, 1e-3, 'c2', None, None, None, T.PARITY_K_TYPICAL_WINOGRAD, tag, 6, 0)
        except AssertionError as e:
            failed = str(e)[:1500]
    rep = json.load(open(os.path.join(T.OUT, 'teacher_forced_%s.json' % tag)))
    rep['variant'] = variant
    rep['shipped_bounds_exceeded'] = failed
    summary = dict(variant=variant, loss_rel_worst=max(l for _, l in rep['loss_rel_per_step']), loss_bar_north_star=1e-3,
                   typical_worst=rep.get('typical_worst', [])[:8], baseline_bimodal_worst=rep.get('baseline_bimodal_worst', [])[:8],
                   events=rep.get('events'), shipped_bounds_exceeded=failed, algo=rep['algo'])
    rep['summary'] = summary
    with open(os.path.join(T.OUT, 'teacher_forced_%s.json' % tag), 'w') as f:
        json.dump(rep, f)
    print(json.dumps(summary, indent=1)[:4000])


if __name__ == '__main__':
    main()
