#!/usr/bin/env python
"""Winograd F(2x2,3x3) forward / data gradient / weight gradient against the direct-form HIP kernels and the fp64 torch
reference at the full-size shapes of the path (diagnostic: prints max-abs errors relative to max|ref|)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F
from neurips18_hierchical_image_manipulation_amd import ops


def run(B, Cin, H, W, Cout, pm, param):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (Cin * 9) ** -0.5
    gy = torch.randn(B, Cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    xp = F.pad(xr, (1, 1, 1, 1), mode='reflect') if pm == 'reflect' else F.pad(xr, (1, 1, 1, 1))
    yr = F.conv2d(xp, wr)
    gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.double())
    out = []
    for wino in (16, 0):
        prev = ops.set_winograd_min_channels(wino)
        xd = x.cuda().requires_grad_(True)
        wd = w.cuda().requires_grad_(True)
        if param:
            wd = torch.nn.Parameter(wd.detach())
        y = ops.conv2d(xd, wd, None, 1, 1, pm, 'none', 0.0)
        gx, gw = torch.autograd.grad(y, (xd, wd), gy.cuda())
        torch.cuda.synchronize()
        ops.set_winograd_min_channels(prev)
        e = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())  # noqa: E731
        out.append((e(y, yr.detach()), e(gx, gxr), e(gw, gwr)))
    print('B%d Cin%d %dx%d Cout%d %-7s param=%d  wino fwd/dgrad/wgrad %.1e %.1e %.1e   direct %.1e %.1e %.1e' % (
        (B, Cin, H, W, Cout, pm, param) + out[0] + out[1]), flush=True)


if __name__ == '__main__':
    for param in (0, 1):
        run(2, 128, 8, 8, 256, 'zero', param)
        run(2, 256, 16, 16, 256, 'zero', param)
        run(2, 512, 16, 16, 512, 'zero', param)
        run(16, 512, 16, 16, 512, 'zero', param)
        run(2, 512, 16, 32, 512, 'reflect', param)
        run(2, 1024, 16, 32, 1024, 'reflect', param)
        run(8, 1024, 16, 32, 1024, 'reflect', param)
