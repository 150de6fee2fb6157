#!/usr/bin/env python
"""Where does a tensor's excess distance from float64 come from?  (development aid for the parity suite)

Runs a few teacher-forced steps of a golden configuration: per step the fp32 oracle and its float64 twin run ONCE, then the
HIP step is repeated from the same adopted state under several kernel selections (HimAlgo variants / schedule switches);
for a handful of tensors the relative L2 distance of every variant's gradient from the float64 gradient is printed next
to the oracle's own.  A variant that brings a tensor down to the oracle's level names the kernel family responsible.

    python tools/parity_probe.py c1_traj 4 [tensor-substring ...]
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch  # noqa: E402


def main():
    import fp64_anchor as fa
    import test_model_gpu as T
    from neurips18_hierchical_image_manipulation_amd import synth, ops, config, _cabi as cb
    tag, steps = sys.argv[1], int(sys.argv[2])
    want = sys.argv[3:] or ['G/model.38.', 'G/model.34.weight', 'G/model.1.weight', 'D/scale0_layer0.0.weight',
                            'D/scale0_layer3.0.weight']
    g = T.load_golden(tag)
    flags = json.loads(str(g['flags']))
    B, H, W = int(g['B']), int(g['H']), int(g['W'])
    variants = [
        ('default', {}, {}),
        ('direct form', dict(wino_min_c=-1), {}),
        ('no fused wino', dict(wino_fused_min_c=-1), {}),
        ('no fewout tiled', dict(disable=cb.ALGO_NO_FEWOUT_TILED), {}),
        ('no fewin tiled', dict(disable=cb.ALGO_NO_FEWIN_TILED), {}),
        ('no fewch mfma', dict(disable=cb.ALGO_NO_FEWCH_MFMA), {}),
        ('no small win', dict(disable=cb.ALGO_NO_SMALL_WIN), {}),
        ('no split-K', dict(disable=cb.ALGO_NO_SPLITK), {}),
        ('generic conv', dict(disable=cb.ALGO_GENERIC_CONV), {}),
        ('no onehot stem', {}, dict(onehot_stem=False)),
        ('D concat input', {}, dict(d_split_input=False)),
        ('no bgemm', dict(disable=cb.ALGO_NO_BGEMM), {}),
    ]
    model, om, om64 = T.build(flags), fa.make_oracle(flags), fa.make_oracle(flags, torch.float64)
    names = None
    table = {v[0]: [] for v in variants}
    table['oracle fp32'] = []
    for s in range(steps):
        fa.adopt64(om64, om)
        state = (om.netG.state_dict(), om.netD.state_dict())
        before = fa.snapshot(om)
        b = synth.make_batch(s, 0, B, H, W, flags.get('label_nc', 35))
        hip = {}
        for name, algo, sched in variants:
            T._adopt(model, om)
            with ops.algo_scope(**dict(T.PINNED_ALGO, **algo)), config.schedule(**sched):
                model.optimize_parameters(b)
                model.sync()
            q = T._hip_quantities(model, before)
            if names is None:
                names = [n for n in q if any(w in n for w in want)]
            hip[name] = {k: q[k]['grad'].detach().double().cpu().clone() for k in names}
        om.optimize_parameters(b)
        fa.step64(om64, b)
        q32, q64 = fa.oracle_quantities(om, before), fa.oracle_quantities(om64, before)
        table['oracle fp32'].append([fa.rel_l2(q32[n]['grad'], q64[n]['grad']) for n in names])
        for name, _, _ in variants:
            table[name].append([fa.rel_l2(hip[name][n], q64[n]['grad']) for n in names])
        print('step %d done' % s, flush=True)
        del state
    print('relative L2 distance of the gradient from the float64 step, per step')
    for i, n in enumerate(names):
        print(n)
        for name in ['oracle fp32'] + [v[0] for v in variants]:
            print('    %-18s %s' % (name, ' '.join('%.2e' % row[i] for row in table[name])))
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'parity_probe_%s.json' % tag), 'w') as f:
        json.dump(dict(tensors=names, table=table), f)


if __name__ == '__main__':
    main()
