"""Do the results of a training step depend on what freed device memory held?  (python tools/poison_probe.py [tag ...])

Every buffer the host side hands to the library comes from ``torch.empty`` (outputs, workspaces): a kernel that reads a
word nobody wrote -- a padded panel column, an unwritten split-K slab row, a table entry past the last run -- computes
with whatever an earlier tensor left there.  In a fresh process that is zeros (the caching allocator's first segments
come zero-filled from the driver); deep inside a test suite it is an old activation.  This tool runs the same seeded
step of a golden toy configuration three times in one process: allocations as they come, every ``torch.empty`` filled
with NaN, and filled with 1e4; the gradients of every parameter must be bit-identical across the three runs."""
import json
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]

_empty, _empty_like = torch.empty, torch.empty_like
FILL = [None]


def _poisoned(*a, **k):
    t = _empty(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(FILL[0])
    return t


def _poisoned_like(*a, **k):
    t = _empty_like(*a, **k)
    if FILL[0] is not None and t.is_cuda and t.is_floating_point():
        t.fill_(FILL[0])
    return t


def one_step(tag, steps=2):
    from test_model_gpu import build
    from util import load_golden
    from neurips18_hierchical_image_manipulation_amd import synth
    g = load_golden(tag)
    flags = g['flags'] if isinstance(g['flags'], dict) else json.loads(str(g['flags']))
    B, H, W = int(g['B']), int(g['H']), int(g['W'])
    color = bool(int(g['color'])) if 'color' in g else False
    model = build(flags)
    out = {}
    for s in range(steps):
        b = synth.make_batch(s, 0, B, H, W, flags.get('label_nc', 35), color)
        ld = model.optimize_parameters(b)
        model.sync()
        for tg, net in (('G', model.netG), ('D', model.netD)):
            for k, p in net.named_parameters():
                if p.grad is not None:
                    out['%d/%s/%s' % (s, tg, k)] = p.grad.detach().clone()
        for k, v in ld.items():
            out['%d/loss/%s' % (s, k)] = v.detach().clone().reshape(-1)
    return out


def main():
    tags = sys.argv[1:] or ['tiny_twostream', 'tiny_global', 'tiny_color']
    torch.empty, torch.empty_like = _poisoned, _poisoned_like
    rc = 0
    for tag in tags:
        runs = {}
        for name, fill in (('plain', None), ('nan', float('nan')), ('1e4', 1e4), ('plain2', None)):
            FILL[0] = fill
            runs[name] = one_step(tag)
            FILL[0] = None
            torch.cuda.synchronize()
        base = runs['plain']
        for name in ('nan', '1e4', 'plain2'):
            bad = []
            for k, v in base.items():
                w = runs[name][k]
                if not torch.equal(v, w):
                    nan = int(torch.isnan(w).sum())
                    d = float((v.double() - w.double()).norm() / max(float(v.double().norm()), 1e-30)) if not nan else float('nan')
                    bad.append((k, nan, d))
            print('%-16s %-7s: %d of %d tensors differ from the plain run' % (tag, name, len(bad), len(base)))
            for k, nan, d in bad[:12]:
                print('      %-60s nan=%d rel=%.3e' % (k, nan, d))
            if bad:
                rc = 1
    return rc


if __name__ == '__main__':
    sys.exit(main())
