#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the hot layer shapes of config C2 (fwd / dgrad / wgrad), TFLOP/s each.
Usage on the GPU box: python tools/conv_bench.py [filter]"""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import ops

SHAPES = [
    # name, B, Cin, H, W, Cout, k, s, p, mode
    ('res3x3_1024', 8, 1024, 16, 32, 1024, 3, 1, 1, 'reflect'),
    ('stem7_38_64', 8, 38, 256, 512, 64, 7, 1, 3, 'reflect'),
    ('down_64_128', 8, 64, 256, 512, 128, 3, 2, 1, 'zero'),
    ('down_512_1024', 8, 512, 32, 64, 1024, 3, 2, 1, 'zero'),
    ('vgg_64_64', 8, 64, 256, 512, 64, 3, 1, 1, 'zero'),
    ('vgg_128_128', 8, 128, 128, 256, 128, 3, 1, 1, 'zero'),
    ('vgg_256_256', 8, 256, 64, 128, 256, 3, 1, 1, 'zero'),
    ('vgg_512_512', 8, 512, 32, 64, 512, 3, 1, 1, 'zero'),
    ('d_256_512_s1', 8, 256, 33, 65, 512, 4, 1, 2, 'zero'),
    ('d_41_64_s2', 8, 41, 256, 512, 64, 4, 2, 2, 'zero'),
    ('d_head', 8, 512, 34, 66, 1, 4, 1, 2, 'zero'),
    ('g_head', 8, 64, 256, 512, 3, 7, 1, 3, 'reflect'),
    ('vgg_3_64', 8, 3, 256, 512, 64, 3, 1, 1, 'zero'),
    ('b2m_stem_70_64', 32, 70, 256, 256, 64, 5, 1, 2, 'zero'),
    ('b2m_d_71_64_s2', 32, 71, 256, 256, 64, 4, 2, 2, 'zero'),
]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    print('%-16s %10s %10s %10s   (TFLOP/s; ms)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
    for name, B, Cin, H, W, Cout, k, s, p, mode in SHAPES:
        if flt and flt not in name:
            continue
        x = torch.randn(B, Cin, H, W, device='cuda').requires_grad_(True)
        w = (torch.randn(Cout, Cin, k, k, device='cuda') * 0.02).requires_grad_(True)
        y = ops.conv2d(x, w, None, s, p, mode, 'none')
        gy = torch.randn_like(y)
        flops = 2.0 * y.numel() * Cin * k * k
        with torch.no_grad():
            t_f = timeit(lambda: ops.conv2d(x, w, None, s, p, mode, 'none'))
        x2 = x.detach().requires_grad_(True)
        w2 = w.detach()
        y2 = ops.conv2d(x2, w2, None, s, p, mode, 'none')
        t_d = timeit(lambda: torch.autograd.grad(y2, x2, gy, retain_graph=True))
        x3 = x.detach()
        w3 = w.detach().requires_grad_(True)
        y3 = ops.conv2d(x3, w3, None, s, p, mode, 'none')
        t_w = timeit(lambda: torch.autograd.grad(y3, w3, gy, retain_graph=True))
        print('%-16s %6.1f %5.2fms %6.1f %5.2fms %6.1f %5.2fms' % (
            name, flops / t_f / 1e9, t_f, flops / t_d / 1e9, t_d, flops / t_w / 1e9, t_w))


if __name__ == '__main__':
    main()
