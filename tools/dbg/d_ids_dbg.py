import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
import torch, torch.nn.functional as F
from neurips18_hierchical_image_manipulation_amd import ops, config
from test_ops_gpu import _label_blocks, _rand
B, NC, Cd, H, W, Cout, block = 8, 35, 3, 256, 512, 64, 16
lab = _label_blocks(B, H, W, NC, block, seed=5)
dense = _rand(B, Cd, H, W, seed=6)
img = _rand(B, 3, H, W, seed=7)
Cin = NC + Cd + 3
w = (_rand(Cout, Cin, 4, 4, seed=8) * 0.05)
b = (_rand(Cout, seed=9) * 0.1)
onehot = torch.zeros(B, NC, H, W)
valid = (lab >= 0) & (lab < NC)
onehot.scatter_(1, lab.clamp(0, NC - 1).long(), valid.float())
xin = torch.cat([onehot, dense, img], 1).double()
w64 = w.double().requires_grad_(True)
y_ref = F.conv2d(xin, w64, b.double(), 2, 2)
gy = _rand(*y_ref.shape, seed=4)
(gw_ref,) = torch.autograd.grad(y_ref, w64, gy.double())
w32 = w.clone().requires_grad_(True)
y32 = F.conv2d(xin.float(), w32, b, 2, 2)
(gw32,) = torch.autograd.grad(y32, w32, gy)
def rel(a, r): return float((a.double().cpu() - r).norm() / r.norm())
print('torch cpu fp32 vs fp64: onehot %.3e dense %.3e' % (rel(gw32[:, :NC], gw_ref[:, :NC]), rel(gw32[:, NC:], gw_ref[:, NC:])))
for from_ids in (True, False):
    with config.schedule(d_from_ids=from_ids):
        cond = ops.LabelCond(lab.cuda(), NC, dense.cuda())
        wd = w.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True); imd = img.cuda().requires_grad_(True)
        y = ops.cond_image_conv2d(cond, imd, wd, bd, 2, 2, 'none', 0.2)
        gw, = torch.autograd.grad(y, wd, gy.cuda())
        print('from_ids', from_ids, 'fwd %.3e' % rel(y, y_ref), 'wgrad onehot %.3e dense-cond %.3e image %.3e' % (
            rel(gw[:, :NC], gw_ref[:, :NC]), rel(gw[:, NC:NC + Cd], gw_ref[:, NC:NC + Cd]), rel(gw[:, NC + Cd:], gw_ref[:, NC + Cd:])))
        d = (gw.double().cpu() - gw_ref).abs()
        i = d.argmax(); print('   worst abs', float(d.max()), 'at', [int(v) for v in torch.unravel_index(i, d.shape)], 'ref', float(gw_ref.flatten()[i]))
# the 6-channel dense conv on its own through the plain op
x6 = torch.cat([dense, img], 1)
w6 = w[:, NC:].contiguous()
w6d = w6.double().requires_grad_(True)
y6 = F.conv2d(x6.double(), w6d, None, 2, 2)
(g6,) = torch.autograd.grad(y6, w6d, gy.double())
wd = w6.cuda().requires_grad_(True)
y = ops.conv2d(x6.cuda(), wd, None, 2, 2, 'zero', 'none')
(gw,) = torch.autograd.grad(y, wd, gy.cuda())
print('plain 6-channel conv: fwd %.3e wgrad %.3e' % (rel(y, y6), rel(gw, g6)))
