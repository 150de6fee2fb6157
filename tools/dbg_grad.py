import sys, json, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_model_gpu as T
from util import load_golden
from neurips18_hierchical_image_manipulation_amd import synth
tag, target = sys.argv[1], int(sys.argv[2])
g = load_golden(tag); flags = json.loads(str(g['flags']))
B,H,W = int(g['B']),int(g['H']),int(g['W'])
model, om = T.build(flags), T._oracle_for(flags)
for s in range(target + 1):
    T._adopt(model, om)
    b = synth.make_batch(s,0,B,H,W,flags.get('label_nc',35))
    got = model.optimize_parameters(b); ref = om.optimize_parameters(b)
rows=[]
for tg,(hnet,onet) in (('G',(model.netG, om.netG)),('D',(model.netD, om.netD))):
    dead = T._biases_in_front_of_instance_norm(hnet)
    for (name,hp),op in zip(hnet.named_parameters(), onet.parameters()):
        if name in dead: continue
        gr = op.grad; scale = gr.double().norm().item(); d=(hp.grad.cpu()-gr).abs(); e=d.double().norm().item()
        nbad = int((d > 1e-4*scale).sum())
        rows.append((e/max(scale,1e-30), tg+'.'+name, scale, e, nbad, d.numel()))
rows.sort(reverse=True)
for r in rows[:12]: print('%.3e %-40s scale %.3e err %.3e  elems>1e-4: %d of %d' % r)
