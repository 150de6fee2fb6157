"""Golden vectors of the spectral-norm LAYERS from the REAL reference classes models/sn_utils.py:28-72.

Build container only:  python tests/golden/make_golden_sn.py  ->  tests/golden/sn_layers.npz
For SNLinear(24, 10) and SNConv2d(6, 10, 3, 1, 1): seeded W / b / u0 / x / gy, the reference's training-mode output, the
persisted u afterwards, and the gradients of sum(y * gy) with respect to W, b and x (through BOTH normalisations of the
power iteration -- the reference detaches nothing).  The oracle's restatement (oracle/ref_cpu.py) is asserted against it."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from oracle import ref_shim, ref_cpu                                     # noqa: E402


def run(layer, W, b, u0, x, gy):
    with torch.no_grad():
        layer.weight.copy_(W)
        layer.bias.copy_(b)
        if isinstance(layer.u, torch.nn.Parameter):
            layer.u.data.copy_(u0)
        else:
            layer.u = u0.clone()
    layer.train()
    xin = x.clone().requires_grad_(True)
    y = layer(xin)
    gW, gb, gx = torch.autograd.grad((y * gy).sum(), [layer.weight, layer.bias, xin])
    return y.detach(), layer.u.detach().clone(), gW, gb, gx


def main():
    _, _, _, S, _ = ref_shim.nets()
    g = torch.Generator().manual_seed(29)
    out = {}
    cases = {
        'lin': (lambda: S.SNLinear(24, 10), lambda: ref_cpu.SNLinear(24, 10), (10, 24), (4, 24), (4, 10)),
        'conv': (lambda: S.SNConv2d(6, 10, 3, 1, 1), lambda: ref_cpu.SNConv2d(6, 10, 3, 1, 1), (10, 6, 3, 3), (2, 6, 9, 11),
                 (2, 10, 9, 11)),
    }
    for tag, (mk_ref, mk_ora, wshape, xshape, yshape) in cases.items():
        W = torch.randn(*wshape, generator=g) * 0.1
        b = torch.randn(wshape[0], generator=g) * 0.1
        u0 = torch.randn(1, wshape[0], generator=g)
        x = torch.randn(*xshape, generator=g)
        gy = torch.randn(*yshape, generator=g)
        r = run(mk_ref(), W, b, u0, x, gy)
        o = run(mk_ora(), W, b, u0, x, gy)
        for a, c, name in zip(r, o, ('y', 'u', 'gW', 'gb', 'gx')):
            assert torch.allclose(a, c, rtol=1e-6, atol=1e-7), (tag, name, (a - c).abs().max())
        out.update({tag + '_W': W.numpy(), tag + '_b': b.numpy(), tag + '_u0': u0.numpy(), tag + '_x': x.numpy(),
                    tag + '_gy': gy.numpy()})
        for a, name in zip(r, ('y', 'u', 'gW', 'gb', 'gx')):
            out['%s_%s' % (tag, name)] = a.numpy()
        print(tag, 'pinned: |y| %.3f  sigma-normalised W max %.3f' % (float(r[0].abs().max()), float(W.abs().max())))
    np.savez_compressed(os.path.join(HERE, 'sn_layers.npz'), **out)


if __name__ == '__main__':
    main()
