"""Time the data gradient of the first PatchGAN conv towards the 3 image channels (tiny-M kernel) at C2: python tools/dimg_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from neurips18_hierchical_image_manipulation_amd import ops
torch.manual_seed(0)
for (H, W) in ((256, 512), (128, 256), (64, 128)):
    x = torch.randn(8, 3, H, W, device='cuda', requires_grad=True)
    w = torch.nn.Parameter(torch.randn(64, 3, 4, 4, device='cuda') * 0.1)
    y = ops.conv2d(x, w, None, 2, 2, 'zero', 'none')
    gy = torch.randn_like(y)
    ref = torch.nn.functional.conv2d(x.detach().cpu().double().requires_grad_(True), w.detach().cpu().double(), None, 2, 2)
    xr = x.detach().cpu().double().requires_grad_(True)
    (gref,) = torch.autograd.grad(torch.nn.functional.conv2d(xr, w.detach().cpu().double(), None, 2, 2), xr, gy.cpu().double())
    for _ in range(3):
        (gx,) = torch.autograd.grad(y, x, gy, retain_graph=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        (gx,) = torch.autograd.grad(y, x, gy, retain_graph=True)
    e1.record()
    torch.cuda.synchronize()
    err = float((gx.cpu().double() - gref).abs().max() / gref.abs().max())
    print('%dx%d: dgrad(+wgrad skipped) %.1f us per call, max rel err %.2e' % (H, W, e0.elapsed_time(e1) / 20 * 1e3, err))
