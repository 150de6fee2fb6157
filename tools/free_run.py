#!/usr/bin/env python
"""Free-running training trajectory of a golden configuration on the HIP path, printed as one JSON line
(per-step max relative loss deviation from tests/golden/<tag>.npz = the REAL reference's trajectory).

    python tools/free_run.py c2_traj [steps] [--algo '{"wino_min_c": -1, "ksplit_max": 8, ...}']

Run as a subprocess by tests/test_model_gpu.py (a fresh process and allocator per 20-step full-size run).  Kernel
selection = the HimAlgo given by --algo (fields of include/him.h HimAlgo; the parity suite pins every field explicitly)
on top of the process default; the resolved HimAlgo and the schedule are part of the output."""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np  # noqa: E402


def main():
    from neurips18_hierchical_image_manipulation_amd import synth, ops, config
    from neurips18_hierchical_image_manipulation_amd.models import create_model
    algo = {}
    if '--algo' in sys.argv:
        i = sys.argv.index('--algo')
        algo = json.loads(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    for k, v in algo.items():
        setattr(ops.current_algo(), k, int(v))
    tag = sys.argv[1]
    g = np.load(os.path.join(ROOT, 'tests', 'golden', tag + '.npz'), allow_pickle=False)
    flags = json.loads(str(g['flags']))
    B, H, W, color = int(g['B']), int(g['H']), int(g['W']), bool(int(g['color']))
    ref = g['losses'].astype(np.float64)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else ref.shape[0]
    names = ['G_GAN', 'G_GAN_Feat', 'G_VGG', 'D_real', 'D_fake']
    model = create_model(dict(flags, gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/him_free_run', name='t'))
    model.netG.load_state_dict(synth.init_state_dict(model.netG.state_dict(), 1))
    model.netD.load_state_dict(synth.init_state_dict(model.netD.state_dict(), 2))
    got = []
    for s in range(steps):
        ld = model.optimize_parameters(synth.make_batch(s, 0, B, H, W, flags.get('label_nc', 35), color))
        got.append([float(ld[k].detach()) for k in names])
    got = np.array(got, np.float64)
    rel = np.abs(got - ref[:steps]) / np.maximum(np.abs(ref[:steps]), 1e-12)
    print('FREE_RUN ' + json.dumps(dict(tag=tag, switches={k: v for k, v in os.environ.items() if k.startswith('HIM_')},
                                       algo=ops.resolved_algo(), schedule=config.SCHED.as_dict(),
                                       rel_per_step=rel.max(axis=1).tolist(), rel=rel.tolist(), got=got.tolist())))


if __name__ == '__main__':
    main()
