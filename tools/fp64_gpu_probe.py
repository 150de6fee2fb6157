"""TEST-INFRASTRUCTURE probe: can the float64 anchor step (tests/fp64_anchor.py: the oracle's own code under
torch.set_default_dtype(float64)) run on the GPU through torch's own double-precision operators, how long does it take at
the parity configurations, and how far is it from the same float64 step on the host?

    python tools/fp64_gpu_probe.py c1_traj [c2_traj ...]      (first tag: also the host float64 step, for the distance)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fp64_anchor as fa      # noqa: E402
from neurips18_hierchical_image_manipulation_amd import synth      # noqa: E402


def main():
    out = {}
    for i, tag in enumerate(sys.argv[1:] or ['c1_traj']):
        g = np.load(os.path.join(ROOT, 'tests', 'golden', tag + '.npz'))
        flags = json.loads(str(g['flags']))
        B, H, W, color = int(g['B']), int(g['H']), int(g['W']), bool(int(g['color']))
        om = fa.make_oracle(flags)
        om64g = fa.make_oracle(flags, torch.float64, device='cuda')
        b = synth.make_batch(0, 0, B, H, W, flags.get('label_nc', 35), color)
        before = fa.snapshot(om)
        fa.adopt64(om64g, om)
        torch.cuda.synchronize()
        t0 = time.time()
        lg = fa.step64(om64g, b)
        torch.cuda.synchronize()
        t_gpu = time.time() - t0
        fa.adopt64(om64g, om)       # second step from the same state: warm timing
        for o in (om64g.optimizer_G, om64g.optimizer_D):
            o.state.clear()
        t0 = time.time()
        lg2 = fa.step64(om64g, b)
        torch.cuda.synchronize()
        t_gpu2 = time.time() - t0
        rec = dict(gpu_first_s=t_gpu, gpu_warm_s=t_gpu2, losses_gpu=lg, losses_gpu_again=lg2,
                   mem_GB=torch.cuda.max_memory_allocated() / 2 ** 30)
        if i == 0:
            om64c = fa.make_oracle(flags, torch.float64)
            fa.adopt64(om64c, om)
            t0 = time.time()
            lc = fa.step64(om64c, b)
            rec['cpu_s'] = time.time() - t0
            rec['losses_cpu'] = lc
            qg, qc = fa.oracle_quantities(om64g, before), fa.oracle_quantities(om64c, before)
            dead = fa.dead_biases(om.netG, 'G') | fa.dead_biases(om.netD, 'D')
            worst = max((fa.rel_l2(qg[n]['grad'], qc[n]['grad']), n) for n in qc if n not in dead)
            rec['worst_grad_gpu64_vs_cpu64'] = worst
            rec['loss_rel_gpu64_vs_cpu64'] = max(abs(lg[k] - lc[k]) / max(abs(lc[k]), 1e-300) for k in lc)
        out[tag] = rec
        print(tag, json.dumps(rec), flush=True)
        del om64g
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'fp64_gpu_probe.json'), 'w') as f:
        json.dump(out, f)


if __name__ == '__main__':
    main()
