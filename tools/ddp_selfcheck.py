#!/usr/bin/env python
"""Run under torch.distributed.run (any world size, nccl): trains 3 steps of a toy model with the bucketed RCCL
reducer attached and checks, for world size 1, bit-identity with the un-attached model (avg over one rank)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.distributed as dist
from neurips18_hierchical_image_manipulation_amd import synth
from neurips18_hierchical_image_manipulation_amd.dist import init_process_group_from_env, attach_data_parallel
from neurips18_hierchical_image_manipulation_amd.models import create_model

os.environ.setdefault('WORLD_SIZE', '1')
rank, local, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ['WORLD_SIZE'])
backend = os.environ.get('HIM_DDP_BACKEND', 'nccl')   # 'gloo': several ranks may share ONE GPU (logic test without RCCL)
local = local % torch.cuda.device_count()
os.environ['LOCAL_RANK'] = str(local)               # models pick their device from LOCAL_RANK
torch.cuda.set_device(local)
if not dist.is_initialized():
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
flags = dict(model='pix2pixHD_condImg', netG='global', ngf=16, ndf=16, n_downsample_global=3, n_blocks_global=2,
             num_D=2, n_layers_D=3, label_nc=35, no_instance=True, gpu_ids=[local], isTrain=True,
             checkpoints_dir='/tmp/him_ddp', name='t')
def build():
    m = create_model(dict(flags))
    m.netG.load_state_dict(synth.init_state_dict(m.netG.state_dict(), 1))
    m.netD.load_state_dict(synth.init_state_dict(m.netD.state_dict(), 2))
    return m
a = build()
attach_data_parallel(a, bucket_bytes=1 << 16, force=True)
assert a.reducer_G is not None and len(a.reducer_G.buckets) > 3
b = build() if world == 1 else None
for s in range(3):
    batch = synth.make_batch(s, rank, 2, 64, 64)
    la = a.optimize_parameters(batch)
    assert all(a.reducer_G.launched) and all(a.reducer_D.launched)
    if b is not None:
        lb = b.optimize_parameters(batch)
        for k in la:
            assert float(la[k].detach()) == float(lb[k].detach()), (k, float(la[k]), float(lb[k]))
torch.cuda.synchronize()
if b is not None:
    for p, q in zip(list(a.netG.parameters()) + list(a.netD.parameters()), list(b.netG.parameters()) + list(b.netD.parameters())):
        assert torch.equal(p, q)
# ranks must hold identical parameters after averaging gradients
chk = torch.stack([p.detach().double().sum() for p in a.netG.parameters()]).sum().reshape(1)
allc = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(allc, chk)
assert all(torch.equal(allc[0], c) for c in allc)
nbG, nbD = len(a.reducer_G.buckets), len(a.reducer_D.buckets)

# ---- the box2mask trainer (BatchNorm generator + 2-scale BatchNorm PatchGAN) through the same reducer -----------------
def build_b2m():
    m = create_model(dict(model='AE_maskgen_twostream', ndf=16, gpu_ids=[local], isTrain=True, checkpoints_dir='/tmp/him_ddp',
                          name='b'))
    m.netG.load_state_dict(synth.init_state_dict(m.netG.state_dict(), 21))
    m.netD.load_state_dict(synth.init_state_dict(m.netD.state_dict(), 22))
    return m
def b2m_step(m, bt):
    out, _ = m.forward(bt['label'], None, bt['mask_ctx_in'], None, bt['mask_out'], bt['mask_obj_inst'], bt['cls'], bt['mask_in'])
    return [float(x.reshape(-1)[0]) if torch.is_tensor(x) else float(x) for x in out]
a = build_b2m()
attach_data_parallel(a, bucket_bytes=1 << 16, force=True)
assert a.reducer_G is not None and a.reducer_D is not None and len(a.reducer_G.buckets) > 3
b = build_b2m() if world == 1 else None
for s in range(3):
    bt = synth.make_box2mask_batch(s, rank, 2, 64, 64)
    la = b2m_step(a, bt)
    assert all(a.reducer_G.launched) and all(a.reducer_D.launched)
    if b is not None:
        assert la == b2m_step(b, bt), (la,)
torch.cuda.synchronize()
if b is not None:
    for p, q in zip(list(a.netG.parameters()) + list(a.netD.parameters()), list(b.netG.parameters()) + list(b.netD.parameters())):
        assert torch.equal(p, q)
chk = torch.stack([p.detach().double().sum() for p in list(a.netG.parameters()) + list(a.netD.parameters())]).sum().reshape(1)
allc = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(allc, chk)
assert all(torch.equal(allc[0], c) for c in allc), 'box2mask replicas diverged'
dist.barrier()
if rank == 0:
    print('DDP SELFCHECK OK world=%d buckets G=%d D=%d (mask2image) G=%d D=%d (box2mask)' % (
        world, nbG, nbD, len(a.reducer_G.buckets), len(a.reducer_D.buckets)))
dist.destroy_process_group()
