import sys, time, torch
sys.path.insert(0, '.')
from oracle import ref_cpu
from neurips18_hierchical_image_manipulation_amd import synth
C1 = dict(model='pix2pixHD_condImg', netG='global', ngf=64, ndf=64, n_downsample_global=4, n_blocks_global=9, num_D=1, n_layers_D=3, label_nc=35, no_instance=True)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**C1))
    om.optimize_parameters(synth.make_batch(0, 0, 1, 128, 256))
    t0 = time.time(); om.optimize_parameters(synth.make_batch(1, 0, 1, 128, 256)); t1 = time.time()
    om.optimize_parameters(synth.make_batch(2, 0, 1, 128, 256)); t2 = time.time()
    print('threads', th, 'C1 s/step %.2f %.2f' % (t1 - t0, t2 - t1), flush=True)
