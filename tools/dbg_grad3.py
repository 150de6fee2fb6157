import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_model_gpu as T
from util import load_golden
from neurips18_hierchical_image_manipulation_amd import synth
from oracle import ref_cpu
tag = sys.argv[1]; extra = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
g = load_golden(tag); flags = dict(json.loads(str(g['flags'])), **extra)
B,H,W = int(g['B']),int(g['H']),int(g['W'])
model = T.build(flags)
om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**flags))
om.netG.load_state_dict(synth.init_state_dict(om.netG.state_dict(), 1)); om.netD.load_state_dict(synth.init_state_dict(om.netD.state_dict(), 2))
if om.vgg is not None: om.vgg.load_state_dict(synth.init_state_dict(om.vgg.state_dict(), 3, 'vgg'))
def g_grads(b):
    losses,_ = model(b['label'], b['inst'], b['image'], None, b['mask_in'], b['mask_out'])
    model.combine_losses(losses); model.optimizer_G.zero_grad(); model.loss_G.backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.netG.parameters()]
for s in range(12):
    T._adopt(model, om)
    b = synth.make_batch(s,0,B,H,W,flags.get('label_nc',35))
    g1 = g_grads(b); g2 = g_grads(b)
    same = all(torch.equal(a,c) for a,c in zip(g1,g2))
    ref = om.optimize_parameters(b)
    dead = T._biases_in_front_of_instance_norm(model.netG); w=0
    for (name,hp),gh,op in zip(model.netG.named_parameters(), g1, om.netG.parameters()):
        if name in dead: continue
        w = max(w, (gh.cpu()-op.grad).abs().max().item()/max(op.grad.abs().max().item(),1e-30))
    print('step', s, 'hip run-to-run identical:', same, ' G grad rel vs oracle %.1e' % w)
    model.optimize_parameters(b)
