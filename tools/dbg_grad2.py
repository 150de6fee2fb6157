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
out=[]
for s in range(20):
    T._adopt(model, om)
    b = synth.make_batch(s,0,B,H,W,flags.get('label_nc',35))
    got = model.optimize_parameters(b); ref = om.optimize_parameters(b)
    worst = {}
    for tg,(hnet,onet) in (('G',(model.netG, om.netG)),('D',(model.netD, om.netD))):
        dead = T._biases_in_front_of_instance_norm(hnet); w=0
        for (name,hp),op in zip(hnet.named_parameters(), onet.parameters()):
            if name in dead: continue
            gr = op.grad; w = max(w, (hp.grad.cpu()-gr).abs().max().item()/max(gr.abs().max().item(),1e-30))
        worst[tg]=w
    out.append(worst)
print(extra, 'G:', ' '.join('%.0e'%o['G'] for o in out)); print('   D:', ' '.join('%.0e'%o['D'] for o in out))
