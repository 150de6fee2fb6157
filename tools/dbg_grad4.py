import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_model_gpu as T
from util import load_golden
from neurips18_hierchical_image_manipulation_amd import synth
from oracle import ref_cpu
tag = sys.argv[1]; extra = json.loads(sys.argv[2]); target = int(sys.argv[3])
g = load_golden(tag); flags = dict(json.loads(str(g['flags'])), **extra)
B,H,W = int(g['B']),int(g['H']),int(g['W'])
model = T.build(flags)
om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**flags))
om.netG.load_state_dict(synth.init_state_dict(om.netG.state_dict(), 1)); om.netD.load_state_dict(synth.init_state_dict(om.netD.state_dict(), 2))
for s in range(target):
    b = synth.make_batch(s,0,B,H,W,flags.get('label_nc',35)); om.optimize_parameters(b)
T._adopt(model, om)
b = synth.make_batch(target,0,B,H,W,flags.get('label_nc',35))
# HIP
losses, fake = model(b['label'], b['inst'], b['image'], None, b['mask_in'], b['mask_out'], infer=True)
model.combine_losses(losses)
(dfake,) = torch.autograd.grad(model.loss_G, fake, retain_graph=True)
# oracle
lo, fo = om.forward(b['label'], b['inst'], b['image'], None, b['mask_in'], b['mask_out'], infer=True)
loss_G = lo[0] + lo[1] + lo[2]
(dfo,) = torch.autograd.grad(loss_G, fo, retain_graph=True)
print('fake fwd err', (fake.detach().cpu()-fo.detach()).abs().max().item(), 'scale', fo.abs().max().item())
d = (dfake.cpu()-dfo).abs(); sc = dfo.abs().max().item()
print('dfake rel err %.3e' % (d.max().item()/sc), 'n bad', int((d>1e-4*sc).sum()), 'of', d.numel())
idx = (d>1e-4*sc).nonzero()
print(idx[:10].tolist())
# D pass-3 activations compare
with torch.no_grad():
    onehot, cond = om.encode_input(b['label'], b['inst'], b['image'], b['mask_in'])
    din = torch.cat((onehot, cond, fo.detach()), 1)
    po = om.netD(din)
    ph = model.netD(din.cuda())
    for i,(a,c) in enumerate(zip(ph,po)):
        for j,(x,y) in enumerate(zip(a,c)):
            e = (x.cpu()-y).abs().max().item(); 
            # count sign disagreements
            sd = int(((x.cpu()>0) != (y>0)).sum()); nz = int((y.abs()<1e-6).sum())
            print('D scale %d layer %d shape %s err %.2e  sign-disagree %d  |y|<1e-6: %d' % (i,j,tuple(y.shape),e,sd,nz))
