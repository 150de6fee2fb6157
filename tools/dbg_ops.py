import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from neurips18_hierchical_image_manipulation_amd import ops
from util import report
def rnd(*s, seed=0, scale=1.0):
    return torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale
def conv_case(B,Cin,H,W,Cout,k,s,p,pm):
    x = rnd(B,Cin,H,W,seed=1).requires_grad_(True); w = rnd(Cout,Cin,k,k,seed=2,scale=(Cin*k*k)**-0.5).requires_grad_(True); b = rnd(Cout,seed=3,scale=0.1).requires_grad_(True)
    xp = F.pad(x,(p,p,p,p),mode='reflect') if pm=='reflect' else x
    y = F.conv2d(xp, w, b, s, 0 if pm=='reflect' else p); gy = rnd(*y.shape, seed=4)
    gx,gw,gb = torch.autograd.grad(y,(x,w,b),gy)
    xd,wd,bd = (t.detach().cuda().requires_grad_(True) for t in (x,w,b))
    yd = ops.conv2d(xd,wd,bd,s,p,pm,'none'); g = torch.autograd.grad(yd,(xd,wd,bd),gy.cuda())
    for n,a,r in (('fwd',yd,y),('dgrad',g[0],gx),('wgrad',g[1],gw),('bgrad',g[2],gb)):
        print('  ', report(n,a,r,5e-5)[1])
def deconv_case(B,Cin,H,W,Cout):
    x = rnd(B,Cin,H,W,seed=1).requires_grad_(True); w = rnd(Cin,Cout,3,3,seed=2,scale=(Cin*9)**-0.5).requires_grad_(True); b = rnd(Cout,seed=3,scale=0.1).requires_grad_(True)
    y = F.conv_transpose2d(x,w,b,stride=2,padding=1,output_padding=1); gy = rnd(*y.shape, seed=4)
    gx,gw,gb = torch.autograd.grad(y,(x,w,b),gy)
    xd,wd,bd = (t.detach().cuda().requires_grad_(True) for t in (x,w,b))
    yd = ops.conv_transpose2d(xd,wd,bd,2,1,1,'none'); g = torch.autograd.grad(yd,(xd,wd,bd),gy.cuda())
    for n,a,r in (('fwd',yd,y),('dgrad',g[0],gx),('wgrad',g[1],gw),('bgrad',g[2],gb)):
        print('  ', report(n,a,r,5e-5)[1])
print('res 1x1024x8x16'); conv_case(1,1024,8,16,1024,3,1,1,'reflect')
print('down 512->1024 16x32'); conv_case(1,512,16,32,1024,3,2,1,'zero')
print('down 64->128 128x256'); conv_case(1,64,128,256,128,3,2,1,'zero')
print('deconv 1024->512 8x16'); deconv_case(1,1024,8,16,512)
print('deconv 128->64 64x128'); deconv_case(1,128,64,128,64)
print('D 256->512 s1 17x33'); conv_case(1,256,17,33,512,4,1,2,'zero')
print('D 64->128 s2 65x129'); conv_case(1,64,65,129,128,4,2,2,'zero')
print('vgg 512 8x16'); conv_case(1,512,8,16,512,3,1,1,'zero')
