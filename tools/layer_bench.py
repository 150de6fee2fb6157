#!/usr/bin/env python
"""One layer of the C2 step on its own (forward, data gradient, weight gradient through the product's own autograd ops),
N repetitions each -- run under `rocprofv3 --kernel-trace` by tools/roofline_table.sh so that every kernel of the layer
shows up with its launch grid and its ISOLATED duration; tools/roofline_table.py joins that with the step trace.

    python tools/layer_bench.py <layer> [reps]          python tools/layer_bench.py --list
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

# name: (kind, B, Cin, H, W, Cout, k, stride, pad, mode, frozen, where it sits in the step)
LAYERS = {
    'g_down1': ('conv', 8, 64, 256, 512, 128, 3, 2, 1, 'zero', False, 'GlobalGenerator down 64->128'),
    'g_down2': ('conv', 8, 128, 128, 256, 256, 3, 2, 1, 'zero', False, 'GlobalGenerator down 128->256'),
    'g_down3': ('conv', 8, 256, 64, 128, 512, 3, 2, 1, 'zero', False, 'GlobalGenerator down 256->512'),
    'g_down4': ('conv', 8, 512, 32, 64, 1024, 3, 2, 1, 'zero', False, 'GlobalGenerator down 512->1024'),
    'g_res': ('conv', 8, 1024, 16, 32, 1024, 3, 1, 1, 'reflect', False, 'ResnetBlock conv (x18)'),
    'g_up1': ('deconv', 8, 1024, 16, 32, 512, 3, 2, 1, 'zero', False, 'GlobalGenerator up 1024->512'),
    'g_up2': ('deconv', 8, 512, 32, 64, 256, 3, 2, 1, 'zero', False, 'GlobalGenerator up 512->256'),
    'g_up3': ('deconv', 8, 256, 64, 128, 128, 3, 2, 1, 'zero', False, 'GlobalGenerator up 256->128'),
    'g_up4': ('deconv', 8, 128, 128, 256, 64, 3, 2, 1, 'zero', False, 'GlobalGenerator up 128->64'),
    'd0_l1': ('conv', 8, 64, 129, 257, 128, 4, 2, 2, 'zero', False, 'PatchGAN scale 0 layer 1 (x3 passes)'),
    'd0_l2': ('conv', 8, 128, 65, 129, 256, 4, 2, 2, 'zero', False, 'PatchGAN scale 0 layer 2'),
    'd0_l3': ('conv', 8, 256, 33, 65, 512, 4, 1, 2, 'zero', False, 'PatchGAN scale 0 layer 3'),
    'd1_l1': ('conv', 8, 64, 65, 129, 128, 4, 2, 2, 'zero', False, 'PatchGAN scale 1 layer 1'),
    'd1_l3': ('conv', 8, 256, 17, 33, 512, 4, 1, 2, 'zero', False, 'PatchGAN scale 1 layer 3'),
    'vgg1_2': ('conv', 8, 64, 256, 512, 64, 3, 1, 1, 'zero', True, 'VGG conv1_2 (fused Winograd)'),
    'vgg2_1': ('conv', 8, 64, 128, 256, 128, 3, 1, 1, 'zero', True, 'VGG conv2_1 (fused Winograd)'),
    'vgg2_2': ('conv', 8, 128, 128, 256, 128, 3, 1, 1, 'zero', True, 'VGG conv2_2 (F(4x4))'),
    'vgg3_1': ('conv', 8, 128, 64, 128, 256, 3, 1, 1, 'zero', True, 'VGG conv3_1 (F(4x4))'),
    'vgg3_2': ('conv', 8, 256, 64, 128, 256, 3, 1, 1, 'zero', True, 'VGG conv3_2..3_4 (F(4x4))'),
    'vgg4_2': ('conv', 8, 512, 32, 64, 512, 3, 1, 1, 'zero', True, 'VGG conv4_2..4_4 (F(4x4))'),
}


def main():
    if len(sys.argv) < 2 or sys.argv[1] == '--list':
        print(' '.join(LAYERS))
        return
    import torch
    from neurips18_hierchical_image_manipulation_amd import ops
    name = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    kind, B, Cin, H, W, Cout, k, s, p, mode, frozen, _ = LAYERS[name]
    dev = 'cuda'
    x = torch.randn(B, Cin, H, W, device=dev).requires_grad_(True)
    shape = (Cout, Cin, k, k) if kind == 'conv' else (Cin, Cout, k, k)
    w = torch.nn.Parameter(torch.randn(shape, device=dev) * 0.02, requires_grad=not frozen)
    if frozen:
        w._him_frozen = True

    def fwd(xx):
        if kind == 'conv':
            return ops.conv2d(xx, w, None, s, p, mode, 'none')
        return ops.conv_transpose2d(xx, w, None, s, p, 1, 'none')

    y = fwd(x)
    gy = torch.randn_like(y)
    for _ in range(reps + 1):
        y = fwd(x)
        if frozen:
            torch.autograd.grad(y, x, gy)
        else:
            w.grad = None
            y.backward(gy)
        torch.cuda.synchronize()
    print('%s done: y %s' % (name, tuple(y.shape)))


if __name__ == '__main__':
    main()
