#!/usr/bin/env python
"""The two launches VERDICT r2 singled out as furthest below their roofline, on their own (target of the PMC passes
behind profiles/r03_pmc_few_channel_wgrads.json):
  * weight gradient of the generator head conv7x7 64 -> 3 @256x512, bs 8 (round 2: wgrad_small_win_kernel, 1.9 ms; round 3:
    wgrad_fewch_mfma_kernel on the matrix cores),
  * weight gradient of the one-hot stem conv7x7 38 -> 64 (onehot_wgrad_kernel + the 3 dense channels' few-channel MFMA pass).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d out1 -- python tools/worst_kernels_bench.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d out2 -- python tools/worst_kernels_bench.py
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import ops, synth


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    b = {k: v.cuda() for k, v in synth.make_batch(0, 0, 8, 256, 512).items()}
    # head: 64 -> 3, reflect pad 3
    x = torch.randn(8, 64, 256, 512, device='cuda')
    w = (torch.randn(3, 64, 7, 7, device='cuda') * 0.02).requires_grad_(True)
    y = ops.conv2d(x, w, None, 1, 3, 'reflect', 'none')
    gy = torch.randn_like(y)
    t = timeit(lambda: torch.autograd.grad(y, (w,), gy, retain_graph=True), iters)
    print('head wgrad 64->3 7x7 @256x512 bs8: %.3f ms  (algorithmic reads 281 MB -> %.0f GB/s; 19.7 GFLOP direct form)' % (
        t, 281.0 / t))
    # stem: [one-hot 35 | dense 3] -> 64
    buf, n_label, n_cond = ops.encode_channels(b['label'], b['inst'], b['image'], b['mask_in'], 35, False)
    ws = (torch.randn(64, 38, 7, 7, device='cuda') * 0.02).requires_grad_(True)
    ys = ops.conv2d(buf, ws, None, 1, 3, 'reflect', 'none')
    gys = torch.randn_like(ys)
    t = timeit(lambda: torch.autograd.grad(ys, (ws,), gys, retain_graph=True), iters)
    print('stem wgrad [35 one-hot | 3] -> 64 7x7 @256x512 bs8 (%s): %.3f ms  (dy 268 MB + ids 4 MB + x 12.6 MB)' % (
        ys.grad_fn.__class__.__name__, t))


if __name__ == '__main__':
    main()
