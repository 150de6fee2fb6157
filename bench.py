#!/usr/bin/env python
"""Benchmark of the mask2image training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one full training step (G forward, 3x multi-scale D forward, LSGAN + feature-matching + VGG losses,
G backward + Adam, D backward + Adam) on one synthetic Cityscapes-shaped batch that is already resident in HBM.
Workload = BASELINE.json configs[1]: 512x256 (NCHW (8,.,256,512)), bs 8 per GPU, GlobalGenerator ngf 64 /
4 downsamples / 9 ResnetBlocks (182.6 M params), 3-scale PatchGAN, VGG19 perceptual loss, fp32 throughout.
Rank 0 prints ONE JSON line (metric = BASELINE.json's images/s; weak scaling: bs 8 per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

C2 = dict(model='pix2pixHD_condImg', netG='global', ngf=64, ndf=64, n_downsample_global=4, n_blocks_global=9,
          num_D=3, n_layers_D=3, label_nc=35, no_instance=True)
H, W, BS = 256, 512, 8
PEAK_F32_MFMA = 157.3  # TFLOP/s, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
G_FWD_GFLOP_PER_IMG = 246.3  # conv + transposed-conv FLOPs of GlobalGenerator at 256x512


def _event_ms(fn, iters):
    """HIP-event timing on the stream the kernels are launched on (torch's current stream)."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def dominant_kernel_roofline(device):
    """The launch that carries most of the step's matrix work: stage 2 of the Winograd F(2x2,3x3) ResnetBlock conv
    (refpad(1) + conv3x3 1024->1024 on (8,1024,16,32)) = ONE batched fp32-MFMA GEMM over the 16 transform positions,
    [16] x (1024 x 1024) x (1024 x 1024 tiles) = 34.36 GFLOP EXECUTED per launch (the direct form of the same conv is
    77.31 GFLOP: Winograd does 2.25x fewer multiplies).  18 forward + 18 data-gradient + 18 weight-gradient launches of
    this shape per step.  `achieved` = executed FLOP / average launch time, HIP events on the launch stream around
    him_winograd_gemm (exactly the kernel the conv launches); MFMA-bound: algorithmic bytes = the three 67.1 MB
    operands read/written once = 201 MB, intensity 171 FLOP/B >> the fp32 ridge of ~25.
    `traffic` = HBM-side bytes per launch from the committed PMC passes (profiles/r01_pmc_dominant_kernel.json:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE), null if that file is absent.
    `conv_launch` times the WHOLE conv (input transform + GEMM + output transform, cached weight panel) and states its
    rate in direct-form-equivalent FLOP, for comparison with a non-Winograd implementation."""
    from neurips18_hierchical_image_manipulation_amd import ops
    from neurips18_hierchical_image_manipulation_amd._cabi import lib
    M = K = N = 1024
    a = torch.randn(16, M, K, device=device) * 0.02
    b = torch.randn(16, K, N, device=device)
    c = torch.empty(16, M, N, device=device)
    st = torch.cuda.current_stream().cuda_stream
    gemm = lambda: lib.him_winograd_gemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, K, N, st)  # noqa: E731
    for _ in range(3):
        gemm()
    ms = _event_ms(gemm, 20)
    flops = 2.0 * 16 * M * K * N
    ach = flops / (ms * 1e-3) / 1e12
    # the whole conv launch, as the trainer runs it (Parameter weight: cached Winograd panel)
    x = torch.randn(BS, 1024, 16, 32, device=device)
    w = torch.nn.Parameter(torch.randn(1024, 1024, 3, 3, device=device) * 0.02, requires_grad=False)
    bias = torch.zeros(1024, device=device)
    with torch.no_grad():
        conv = lambda: ops.conv2d(x, w, bias, 1, 1, 'reflect', 'none')  # noqa: E731
        for _ in range(3):
            conv()
        cms = _event_ms(conv, 20)
    direct = 2.0 * 1024 * (BS * 16 * 32) * (1024 * 9)
    traffic = None
    pmc = os.path.join(ROOT, 'profiles', 'r01_pmc_dominant_kernel.json')
    if os.path.isfile(pmc):
        with open(pmc) as f:
            traffic = int(json.load(f)['traffic_bytes_corrected'])
    return dict(bound='mfma',
                kernel='gconv_fast_kernel<2,2,2,2,0,false> as the batched Winograd GEMM [16]x(1024x1024)x(1024x1024) '
                       '(ResnetBlock conv3x3 1024->1024 @16x32, bs 8)',
                achieved=round(ach, 2), peak=PEAK_F32_MFMA, unit='TFLOP/s', frac=round(ach / PEAK_F32_MFMA, 4),
                traffic=traffic, traffic_unit='bytes/launch (PMC, corrected)', algorithmic_bytes=3 * 16 * M * N * 4,
                flop_per_launch=flops, avg_launch_ms=round(ms, 4),
                conv_launch=dict(ms=round(cms, 4), direct_form_gflop=round(direct / 1e9, 2),
                                 direct_form_equivalent_tflops=round(direct / (cms * 1e-3) / 1e12, 1),
                                 executed_tflops=round(flops / (cms * 1e-3) / 1e12, 1)))


def g_forward_roofline(model, batch):
    """The 'fused G-conv forward' the north star prices: whole GlobalGenerator forward at C2.  SURVEY 8(d) counts it in
    direct-form FLOP (1.970 TFLOP per bs-8 batch); the 18 ResnetBlock convs run as Winograd (34.36 instead of 77.31 GFLOP
    each) and the stem's 35 one-hot input channels are table lookups, i.e. 0.967 TFLOP are actually issued to the matrix
    pipe.  Both rates are reported; the roofline fraction is the EXECUTED one."""
    with torch.no_grad():
        model.encode_input(batch['label'], batch['inst'], batch['image'], None, mask_in=batch['mask_in'])
        buf, _, _, mask = model._enc
        fn = lambda: model.netG(buf, mask)  # noqa: E731
        for _ in range(2):
            fn()
        ms = _event_ms(fn, 5)
    direct = G_FWD_GFLOP_PER_IMG * BS / 1e3
    from neurips18_hierchical_image_manipulation_amd import ops
    wino = ops.set_winograd_min_channels(0)
    ops.set_winograd_min_channels(wino)
    executed = direct
    if 0 < wino <= 1024:            # 18 ResnetBlock convs: 34.36 instead of 77.31 GFLOP each
        executed -= 18 * (77.309 - 34.360) / 1e3
    if ops._ONEHOT_ON:              # stem conv7x7 38->64: the 35 one-hot channels are LDS lookups, 3 dense ones stay
        executed -= 2.0 * 64 * 35 * 49 * (BS * H * W) / 1e12
    return dict(ms=round(ms, 3), tflops_direct_form_equivalent=round(direct / (ms * 1e-3), 2),
                tflops_executed=round(executed / (ms * 1e-3), 2),
                frac_of_f32_mfma_peak=round(executed / (ms * 1e-3) / PEAK_F32_MFMA, 4),
                frac_direct_form_equivalent=round(direct / (ms * 1e-3) / PEAK_F32_MFMA, 4))


def cpu_baseline():
    """The CPU oracle (validated bit-exact against the imported reference) timed on this box's host cores on a
    bounded sample of the same workload: full training steps at 512x256 with batch 2 (not 8): one untimed warm-up
    step (oneDNN primitive creation), then one timed step."""
    from oracle import ref_cpu
    from neurips18_hierchical_image_manipulation_amd import synth
    cores = torch.get_num_threads()
    bs = 2
    om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**C2))
    om.optimize_parameters(synth.make_batch(0, 0, bs, H, W))
    b = synth.make_batch(1, 0, bs, H, W)
    t0 = time.time()
    om.optimize_parameters(b)
    dt = time.time() - t0
    return dict(value=round(bs / dt, 4), unit='images/s', cores=cores, kind='port',
                sample='1 timed full training step after 1 warm-up step, 512x256, batch %d of the bs-8 workload, '
                       'torch CPU fp32 oracle, %d threads: %.1f s' % (bs, cores, dt))


C4 = dict(model='pix2pixHD_condImgColor', netG='global_twostream', ngf=64, ndf=64, n_downsample_global=4,
          n_blocks_global=9, num_D=2, n_layers_D=3, label_nc=49, no_instance=True, no_imgCond=True,
          which_encoder='ctx_label', use_skip=True, use_output_gate=True, mask_gan_input=True)


def other_workload(args):
    """Same protocol (resident synthetic batches, barrier + synchronize on both sides, max over ranks) for BASELINE
    config 4 (two-stream colour generator, 256x256, bs 16 per GPU) and config 5's shape (box2mask, 256x256, bs 32)."""
    from neurips18_hierchical_image_manipulation_amd import synth
    from neurips18_hierchical_image_manipulation_amd.dist import init_process_group_from_env, attach_data_parallel
    from neurips18_hierchical_image_manipulation_amd.models import create_model
    rank, local, world = init_process_group_from_env()
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if args.workload == 'c4':
        bs, name = 16, 'C4: mask2image ADE20K-shaped 256x256, colour two-stream generator, label_nc 49, 2-scale PatchGAN + VGG19'
        model = create_model(dict(C4, gpu_ids=[local], isTrain=True, checkpoints_dir='/tmp/him_bench', name='c4', batchSize=bs))
        batches = [{k: v.to(device) for k, v in synth.make_batch(s, rank, bs, 256, 256, 49, True).items()} for s in range(4)]
        step = lambda i: model.optimize_parameters(batches[i % 4])  # noqa: E731
    else:
        bs, name = 32, 'box2mask 256x256 (scripts/train_box2mask_city.sh flags): BatchNorm two-stream mask generator + 2-scale PatchGAN'
        model = create_model(dict(model='AE_maskgen_twostream', gpu_ids=[local], isTrain=True, checkpoints_dir='/tmp/him_bench',
                                  name='b2m'))
        batches = [{k: (v.to(device) if k != 'cls' else v) for k, v in synth.make_box2mask_batch(s, rank, bs, 256, 256).items()}
                   for s in range(4)]

        def step(i):
            b = batches[i % 4]
            return model.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'],
                                 b['mask_in'])[0]
    attach_data_parallel(model)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': '%s train images/sec' % args.workload, 'value': round(bs * world * args.steps / dt, 3),
                          'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
                          'config': {'workload': name, 'global_batch': bs * world, 'per_gpu_batch': bs,
                                     'parallelism': 'dp%d' % world}, 'roofline': None, 'cpu_baseline': None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--workload', choices=['c2', 'c4', 'box2mask'], default='c2',
                    help='c2 (default) = the BASELINE.json metric; c4 / box2mask = the other measured configurations '
                         '(DESIGN.md), reported with the same protocol but without roofline / cpu_baseline legs')
    args = ap.parse_args()
    if args.workload != 'c2':
        return other_workload(args)

    from neurips18_hierchical_image_manipulation_amd import synth
    from neurips18_hierchical_image_manipulation_amd.dist import init_process_group_from_env, attach_data_parallel
    from neurips18_hierchical_image_manipulation_amd.models import create_model

    rank, local, world = init_process_group_from_env()
    if world != max(args.gpus, 1) and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    model = create_model(dict(C2, gpu_ids=[local], isTrain=True, checkpoints_dir='/tmp/him_bench', name='bench',
                              batchSize=BS))
    model.netG.load_state_dict(synth.init_state_dict(model.netG.state_dict(), 1))
    model.netD.load_state_dict(synth.init_state_dict(model.netD.state_dict(), 2))
    attach_data_parallel(model)

    # synthetic batches, resident in HBM before the timed region (4 distinct batches per rank, cycled)
    batches = []
    for s in range(4):
        b = synth.make_batch(s, rank, BS, H, W)
        batches.append({k: v.to(device) for k, v in b.items()})

    def step(i):
        return model.optimize_parameters(batches[i % len(batches)])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            'metric': 'mask2image train images/sec at 512x256 bs=8',
            'value': round(BS * world * args.steps / dt, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'C2: mask2image Cityscapes-shaped 512x256, GlobalGenerator ngf64/4down/9blocks '
                                   '(182.6M params) + 3-scale PatchGAN + VGG19 loss (synthetic weights), full '
                                   'train step G+D Adam, fp32', 'global_batch': BS * world, 'per_gpu_batch': BS,
                       'parallelism': 'dp%d' % world},
            'last_losses': {k: round(float(v.detach()), 5) for k, v in losses.items()},
        }
        if not args.no_roofline:
            out['roofline'] = dominant_kernel_roofline(device)
            out['g_forward'] = g_forward_roofline(model, batches[0])
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
