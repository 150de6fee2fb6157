#!/usr/bin/env python
"""Benchmark of the mask2image training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # N > 1: spawns N ranks itself (one process per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          # same thing, ranks provided by the launcher

One "step" = one full training step (G forward, 3x multi-scale D forward, LSGAN + feature-matching + VGG losses,
G backward + Adam, D backward + Adam) on one synthetic Cityscapes-shaped batch that is already resident in HBM.
Default workload = BASELINE.json configs[1] (C2): 512x256 (NCHW (8,.,256,512)), bs 8 per GPU, GlobalGenerator ngf 64 /
4 downsamples / 9 ResnetBlocks (182.6 M params), 3-scale PatchGAN, VGG19 perceptual loss, fp32 throughout.
Rank 0 prints ONE JSON line (metric = BASELINE.json's images/s; weak scaling: bs 8 per GPU).  `--workload c2local | c4
| box2mask` run the other measured configurations under the same protocol (same JSON schema incl. roofline/cpu_baseline).
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

PEAK_F32_MFMA = 157.3  # TFLOP/s, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)

C2 = dict(model='pix2pixHD_condImg', netG='global', ngf=64, ndf=64, n_downsample_global=4, n_blocks_global=9,
          num_D=3, n_layers_D=3, label_nc=35, no_instance=True)
C1 = dict(C2, num_D=1)                      # BASELINE configs[0]: 256x128 bs 1, global G, 1-scale D (CPU plumbing case)
C2LOCAL = dict(C2, netG='local', ngf=32, n_local_enhancers=1, n_blocks_local=3)
C4 = dict(model='pix2pixHD_condImgColor', netG='global_twostream', ngf=64, ndf=64, n_downsample_global=4,
          n_blocks_global=9, num_D=2, n_layers_D=3, label_nc=49, no_instance=True, no_imgCond=True,
          which_encoder='ctx_label', use_skip=True, use_output_gate=True, mask_gan_input=True)

WORKLOADS = {
    # name: flags, per-GPU batch, H, W, label_nc, colour batch keys, G-forward direct-form GFLOP / image
    'c2': dict(flags=C2, bs=8, H=256, W=512, label_nc=35, color=False, g_gflop=246.3,
               metric='mask2image train images/sec at 512x256 bs=8',
               desc='C2: mask2image Cityscapes-shaped 512x256, GlobalGenerator ngf64/4down/9blocks (182.6M params) + '
                    '3-scale PatchGAN + VGG19 loss (synthetic weights), full train step G+D Adam, fp32'),
    'c2local': dict(flags=C2LOCAL, bs=8, H=256, W=512, label_nc=35, color=False, g_gflop=94.7,
                    metric='mask2image (LocalEnhancer) train images/sec at 512x256 bs=8',
                    desc='C2 with netG=local: LocalEnhancer ngf32 (global G ngf64 at 256x128 + 1 local enhancer, 182.9M '
                         'params) + 3-scale PatchGAN + VGG19 loss (synthetic weights), full train step, fp32'),
    'c4': dict(flags=C4, bs=16, H=256, W=256, label_nc=49, color=True, g_gflop=147.1,
               metric='mask2image colour two-stream train images/sec at 256x256 bs=16',
               desc='C4: mask2image ADE20K-shaped 256x256, pix2pixHD_condImgColor two-stream generator + skips + gate, '
                    'label_nc 49, 2-scale PatchGAN + VGG19 loss, full train step, fp32'),
    'box2mask': dict(flags=dict(model='AE_maskgen_twostream'), bs=32, H=256, W=256, label_nc=35, color=False,
                     g_gflop=23.4, metric='box2mask train images/sec at 256x256 bs=32',
                     desc='C5 shape: box2mask 256x256 (scripts/train_box2mask_city.sh flags), BatchNorm two-stream mask '
                          'generator (13.9M params) + 2-scale BatchNorm PatchGAN, G+D Adam inside forward, fp32'),
}


# --------------------------------------------------------------------------------------------------------------------
# launcher
# --------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def spawn_ranks(args):
    """``python bench.py --gpus N`` from a bare shell: re-execute this script as N ranks (one process per GPU) through
    torch.distributed.run on 127.0.0.1; the ranks' stdout/stderr pass through, so rank 0's JSON line is this process's
    JSON line.  Under torch.distributed.run (RANK set) this is never reached."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', str(max((os.cpu_count() or 8) // args.gpus, 1)))
    return subprocess.call(cmd, env=env)


def check_world(args, world):
    """--gpus N must mean N ranks over RCCL: anything else would print a number for a job that was not run."""
    if world != max(args.gpus, 1):
        raise SystemExit('bench.py: --gpus %d but %d rank(s) are running (WORLD_SIZE)' % (args.gpus, world))
    if world == 1:
        return None
    be = dist.get_backend()
    if be != 'nccl' and os.environ.get('HIM_DDP_BACKEND') != be:      # HIM_DDP_BACKEND=gloo: logic test, ranks share a GPU
        raise SystemExit('bench.py: %d ranks need the RCCL backend ("nccl"), got %r' % (world, be))
    if be == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit('bench.py: --gpus %d but only %d device(s) visible' % (world, torch.cuda.device_count()))
    if dist.get_world_size() != world:
        raise SystemExit('bench.py: process group has %d ranks, expected %d' % (dist.get_world_size(), world))
    return be


# --------------------------------------------------------------------------------------------------------------------
# roofline legs
# --------------------------------------------------------------------------------------------------------------------
def _event_ms(fn, iters):
    """HIP-event timing on the stream the kernels are launched on (torch's current stream)."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def _profile_json(suffix):
    """Newest committed ``profiles/r<NN>_<suffix>`` (rocprofv3 summaries folded by tools/collect_profiles.sh) -> (name, dict)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_' + suffix)))
    if not hits:
        return None, None
    with open(hits[-1]) as f:
        return os.path.join('profiles', os.path.basename(hits[-1])), json.load(f)


def _event_ms_isolated(fn, iters):
    """Average duration of ONE launch: an event pair around every launch, the device drained in between (launches
    issued back to back overlap the tail of one with the ramp-up of the next and under-state the launch by 3-4 %)."""
    tot = 0.0
    for _ in range(iters):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        tot += a.elapsed_time(b)
    return tot / iters


def dominant_kernel_roofline(device, bs, ntiles):
    """The launch that carries most of the step's matrix work: stage 2 of the Winograd F(2x2,3x3) ResnetBlock conv
    (refpad(1) + conv3x3 1024->1024) = ONE batched fp32-MFMA GEMM over the 16 transform positions,
    [16] x (1024 x 1024) x (1024 x ntiles) with ntiles = B * H/2 * W/2 (C2: 8x8x16 = 1024, 34.36 GFLOP EXECUTED per
    launch; the direct form of the same conv is 77.31 GFLOP: Winograd does 2.25x fewer multiplies).  18 forward + 18
    data-gradient + 18 weight-gradient launches of this shape per step.
    `achieved` = executed FLOP / `avg_launch_ms`, the average over 20 launches issued back to back through the C ABI
    (him_winograd_gemm = exactly the kernel / grid the conv launches) between two HIP events on the launch stream -- the
    protocol of the committed rocprofv3 trace (round 3's `frac` used one launch per event pair with the device drained in
    between, 3-4 % more flattering; that figure stays as `avg_launch_ms_single_drained`).  `source: "microbench"`: these are NOT the launches inside the timed training step, which
    share the GPU with other streams; `rocprof` quotes the committed rocprofv3 kernel trace (profiles/
    r<NN>_dominant_kernel_rocprof.json, written by tools/collect_profiles.sh): the same launch isolated -- which
    `avg_launch_ms` must agree with -- and its average inside the traced step.  MFMA-bound: algorithmic bytes = the three
    operands read / written once, intensity 171 FLOP/B >> the fp32 ridge of ~25.
    `traffic` = HBM-side bytes per launch from the committed PMC passes named in `traffic_source` (FETCH_SIZE x2 (gfx950
    correction) + WRITE_SIZE), null if that file is absent or the shape differs -- not measured in this run.
    `conv_launch` times the WHOLE conv (transforms + GEMM, cached weight panel) and states its rate in
    direct-form-equivalent FLOP, for comparison with a non-Winograd implementation."""
    from neurips18_hierchical_image_manipulation_amd import ops, config  # noqa: F401
    from neurips18_hierchical_image_manipulation_amd._cabi import lib
    M = K = 1024
    N = ntiles
    a = torch.randn(16, M, K, device=device) * 0.02
    b = torch.randn(16, K, N, device=device)
    c = torch.empty(16, M, N, device=device)
    st = torch.cuda.current_stream().cuda_stream
    gemm = lambda: lib.him_winograd_gemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, K, N, None, st)  # noqa: E731
    for _ in range(3):
        gemm()
    ms = _event_ms(gemm, 20)              # 20 launches back to back between ONE event pair: what a kernel trace averages
    ms_single = _event_ms_isolated(gemm, 20)
    flops = 2.0 * 16 * M * K * N
    ach = flops / (ms * 1e-3) / 1e12
    # the whole conv launch, as the trainer runs it (Parameter weight: cached Winograd panel)
    hw = ntiles * 4 // bs
    h = 1
    while h * h * 2 < hw:
        h *= 2
    x = torch.randn(bs, 1024, h, hw // h, device=device)
    w = torch.nn.Parameter(torch.randn(1024, 1024, 3, 3, device=device) * 0.02, requires_grad=False)
    bias = torch.zeros(1024, device=device)
    with torch.no_grad():
        conv = lambda: ops.conv2d(x, w, bias, 1, 1, 'reflect', 'none')  # noqa: E731
        for _ in range(3):
            conv()
        cms = _event_ms(conv, 20)
    direct = 2.0 * 1024 * (bs * hw) * (1024 * 9)
    traffic = traffic_src = None
    pmc_name, pmc = _profile_json('pmc_dominant_kernel.json')
    if pmc is not None and int(pmc.get('N', 1024)) == N:
        traffic, traffic_src = int(pmc['traffic_bytes_corrected']), pmc_name
    kt_name, kt = _profile_json('dominant_kernel_rocprof.json')
    rocprof = None
    if kt is not None and int(kt.get('N', 1024)) == N:
        rocprof = dict(file=kt_name, avg_launch_ms_isolated=kt.get('avg_launch_ms'),
                       avg_launch_ms_in_step=kt.get('in_step_avg_launch_ms'),
                       frac_isolated=kt.get('frac_of_f32_mfma_peak'), frac_in_step=kt.get('in_step_frac_of_f32_mfma_peak'))
    # `frac`: the figure a reader can re-derive from the COMMITTED rocprofv3 kernel trace (per-kernel begin -> end durations
    # of the same launch, tools/gemm_bench.py under rocprofv3 --kernel-trace --stats) -- unless THIS run measures the launch
    # slower (round 6, ADVICE r5: a kernel regression must move the headline): frac = min(committed trace, live HIP events).
    # The live figure (`frac_hip_events`) is normally ~5 % higher: back-to-back launches overlap one kernel's tail with the
    # next one's ramp, which an event pair around 20 launches does not count and per-kernel durations count twice.
    ach_events = ach
    frac_source = 'hip_events (no committed rocprofv3 trace of this shape)'
    in_step = None
    if rocprof is not None and rocprof.get('avg_launch_ms_isolated'):
        ach_trace = flops / (float(rocprof['avg_launch_ms_isolated']) * 1e-3) / 1e12
        if ach_trace <= ach_events:
            ach, frac_source = ach_trace, 'rocprofv3 kernel trace: ' + rocprof['file']
        else:
            frac_source = ('hip_events of THIS run: slower than the committed rocprofv3 trace %s (%.4f ms)'
                           % (rocprof['file'], float(rocprof['avg_launch_ms_isolated'])))
        # the same launch INSIDE the traced training step, per GEMM layout (forward, data gradient on the
        # transposed forward panel, weight gradient on the kept input transform; the two backward GEMMs of a layer run
        # concurrently on two streams, `union` = wall-clock the chip spends per GEMM)
        byk = kt.get('in_step_avg_launch_ms_by_kernel') or {}
        names = {'<0, 1>': 'forward', '<1, 1>': 'data_gradient', '<0, 0>': 'weight_gradient'}
        in_step = {names[k2]: dict(avg_launch_ms=v, frac=round(flops / (v * 1e-3) / 1e12 / PEAK_F32_MFMA, 4))
                   for k, v in byk.items() for k2 in names if k2 in k}
        if kt.get('in_step_union_ms_per_launch'):
            in_step['union_of_concurrent_launches'] = dict(ms_per_launch=kt['in_step_union_ms_per_launch'],
                                                           frac=kt.get('in_step_union_frac_of_f32_mfma_peak'))
        in_step['source'] = kt.get('in_step_source', rocprof['file'])
    return dict(bound='mfma', source='microbench',
                kernel='batched Winograd GEMM [16]x(1024x1024)x(1024x%d), bgemm_kernel (fp32 MFMA, LDS-DMA operands; '
                       'ResnetBlock conv3x3 1024->1024, bs %d)' % (N, bs),
                achieved=round(ach, 2), peak=PEAK_F32_MFMA, unit='TFLOP/s', frac=round(ach / PEAK_F32_MFMA, 4),
                frac_source=frac_source,
                achieved_hip_events=round(ach_events, 2), frac_hip_events=round(ach_events / PEAK_F32_MFMA, 4),
                in_step=in_step,
                frac_is='achieved / peak with achieved = flop_per_launch / max(average per-kernel duration of this launch in '
                        'the committed rocprofv3 --kernel-trace of tools/gemm_bench.py = rocprof.avg_launch_ms_isolated, '
                        'avg_launch_ms of this run): the slower of the committed trace and the live measurement; '
                        'frac_hip_events = the same launch measured LIVE in this run: avg_launch_ms = 20 launches back to '
                        'back between two HIP events on the launch stream; avg_launch_ms_single_drained = one launch per '
                        'event pair with the device drained in between; rocprof.*_in_step = the same launches inside the '
                        'traced training step',

                traffic=traffic, traffic_unit='bytes/launch (PMC, corrected)', traffic_source=traffic_src,
                algorithmic_bytes=16 * 4 * (M * K + K * N + M * N),
                flop_per_launch=flops, avg_launch_ms=round(ms, 4), avg_launch_ms_single_drained=round(ms_single, 4),
                frac_single_drained=round(flops / (ms_single * 1e-3) / 1e12 / PEAK_F32_MFMA, 4),
                rocprof=rocprof,
                conv_launch=dict(ms=round(cms, 4), direct_form_gflop=round(direct / 1e9, 2),
                                 direct_form_equivalent_tflops=round(direct / (cms * 1e-3) / 1e12, 1),
                                 executed_tflops=round(flops / (cms * 1e-3) / 1e12, 1)))


def step_flop_accounting(ms_per_step, bs, variant=None):
    """Whole-step roofline of the C2 workload: SURVEY 8(d)'s direct-form count (3 F_G + 9 F_D + 3 F_V = 1225 GFLOP per
    image) minus what this build does not execute, over the measured step time."""
    f_g, f_d, f_v = 246.3, 22.4, 94.7                        # GFLOP per image, forward, direct form
    direct = (3 * f_g + 9 * f_d + 3 * f_v) * bs / 1e3        # TFLOP per step
    # frozen VGG layers on F(4x4,3x3) (1/4 instead of 1/2.25 of the direct-form multiplies): conv3_2..3_4, conv4_1..4_4, conv5_1
    # = 65.2 of VGG's 94.7 GFLOP per image from wino4_min_c = 256 channels; + conv2_2 (9.66) and conv3_1 (4.83) from 128
    from neurips18_hierchical_image_manipulation_amd import ops as _ops
    w4 = _ops.resolved_algo()['wino4_min_c']
    f4 = 0.0 if w4 < 0 else (65.2 + (9.66 + 4.83 if w4 <= 128 else 0.0) if w4 <= 256 else 0.0)
    terms = {
        'winograd_resnet_stack_54_launches': 54 * (77.309 - 34.360) / 1e3,            # F(2x2,3x3): 2.25x fewer multiplies
        'winograd_vgg_3_passes': 3 * (f_v - 0.45) * bs / 1e3 * (1 - 1 / 2.25),        # every VGG conv but conv1_1
        'winograd_f4x4_frozen_vgg_layers': 3 * f4 * bs / 1e3 * (1 / 2.25 - 1 / 4.0),
        'stem_from_label_ids_fwd_and_wgrad': 2 * 2.0 * 64 * 35 * 49 * (bs * 256 * 512) / 1e12,
        'discriminator_passes_7_instead_of_9': 2 * f_d * bs / 1e3,   # shared fake pass; no D weight gradients in loss_G
    }
    from neurips18_hierchical_image_manipulation_amd import config
    if config.SCHED.d_from_ids and config.SCHED.label_ids:
        # round 5: first PatchGAN conv of scale 0 (4x4 s2, 41 -> 64 @ 129x257) reads the 35 one-hot channels as table lookups
        # / run-length sums: 2 forward passes (real, shared fake) + 2 weight gradients
        terms['d_scale0_first_conv_from_label_ids'] = 4 * 2.0 * 64 * 35 * 16 * (bs * 129 * 257) / 1e12
    if variant == 'f4x4-resblock-fwd':
        terms['winograd_f4x4_resblock_forward_18_launches'] = 18 * (34.360 - 19.327) / 1e3
    executed = direct - sum(terms.values())
    return dict(direct_form_tflop=round(direct, 3), not_executed_tflop={k: round(v, 3) for k, v in terms.items()},
                step_executed_tflop=round(executed, 3),
                step_executed_tflops_rate=round(executed / (ms_per_step * 1e-3), 2),
                step_frac_of_peak=round(executed / (ms_per_step * 1e-3) / PEAK_F32_MFMA, 4))


def direct_conv_roofline(device, bs, cin, cout, k, stride, pad, h, w, what):
    """Roofline of a direct-form MFMA conv launch (workloads whose dominant kernel is not the Winograd GEMM)."""
    from neurips18_hierchical_image_manipulation_amd import ops, config  # noqa: F401
    x = torch.randn(bs, cin, h, w, device=device)
    wt = torch.nn.Parameter(torch.randn(cout, cin, k, k, device=device) * 0.02, requires_grad=False)
    bias = torch.zeros(cout, device=device)
    with torch.no_grad():
        conv = lambda: ops.conv2d(x, wt, bias, stride, pad, 'zero', 'none')  # noqa: E731
        for _ in range(3):
            y = conv()
        ms = _event_ms(conv, 20)
    flops = 2.0 * y.numel() * cin * k * k
    ach = flops / (ms * 1e-3) / 1e12
    return dict(bound='mfma', source='microbench', kernel=what, achieved=round(ach, 2), peak=PEAK_F32_MFMA,
                unit='TFLOP/s', frac=round(ach / PEAK_F32_MFMA, 4), traffic=None,
                algorithmic_bytes=4 * (x.numel() + wt.numel() + y.numel()), flop_per_launch=flops,
                avg_launch_ms=round(ms, 4))


def g_forward_roofline(model, batch, wl):
    """The 'fused G-conv forward' the north star prices: the whole generator forward.  SURVEY 8(d) counts it in
    direct-form FLOP; the >= 512-channel 3x3 stride-1 convs run as Winograd (2.25x fewer multiplies) and the stem's
    one-hot input channels are table lookups.  Both rates are reported; the roofline fraction is the EXECUTED one
    (only computed for the C2 GlobalGenerator, whose layer table is fixed: 18 ResnetBlock convs + the 38->64 stem)."""
    from neurips18_hierchical_image_manipulation_amd import ops, config  # noqa: F401
    kw = dict(mask_in=batch['mask_in'])
    if 'obj_mask' in batch:
        kw['obj_mask'] = batch['obj_mask']
    with torch.no_grad():
        input_mask, _, _, _, cond_image = model.encode_input(batch['label'], batch['inst'], batch['image'], None, lazy=True,
                                                             **kw)
        buf, _, _, mask = model._enc
        fn = lambda: model._generate(buf, input_mask, cond_image, mask)  # noqa: E731
        for _ in range(2):
            fn()
        ms = _event_ms(fn, 5)
    bs, H, W = wl['bs'], wl['H'], wl['W']
    direct = wl['g_gflop'] * bs / 1e3
    out = dict(ms=round(ms, 3), direct_form_tflop=round(direct, 4),
               tflops_direct_form_equivalent=round(direct / (ms * 1e-3), 2),
               frac_direct_form_equivalent=round(direct / (ms * 1e-3) / PEAK_F32_MFMA, 4))
    if wl is WORKLOADS['c2']:
        wino = ops.resolved_algo()['wino_min_c']
        executed = direct
        if 0 < wino <= 1024:            # 18 ResnetBlock convs: 34.36 instead of 77.31 GFLOP each
            executed -= 18 * (77.309 - 34.360) / 1e3
            if ops.resolved_algo()['disable'] & (1 << 13):      # --variant f4x4-resblock-fwd: 19.33 GFLOP each
                executed -= 18 * (34.360 - 19.327) / 1e3
        if config.SCHED.onehot_stem:              # stem conv7x7 38->64: the 35 one-hot channels are LDS lookups, 3 dense ones stay
            executed -= 2.0 * 64 * 35 * 49 * (bs * H * W) / 1e12
        out.update(tflops_executed=round(executed / (ms * 1e-3), 2),
                   frac_of_f32_mfma_peak=round(executed / (ms * 1e-3) / PEAK_F32_MFMA, 4))
    return out


# --------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle; test infrastructure, timed AFTER the GPU region, rank 0 at N = 1 only)
# --------------------------------------------------------------------------------------------------------------------
def _cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _time_oracle_steps(make_model, make_batch, step_fn, timed, budget_s):
    om = make_model()
    step_fn(om, make_batch(0))                      # warm-up (oneDNN primitive creation, allocator)
    times = []
    for s in range(timed):
        b = make_batch(1 + s)
        t0 = time.time()
        step_fn(om, b)
        times.append(time.time() - t0)
        if sum(times) > budget_s:
            break
    return times


def cpu_baseline(name, wl, timed=3, budget_s=75.0):
    """The CPU oracle (validated bit-exact against the imported reference, tests/golden/make_golden.py) on this box's
    host cores, the SAME workload at the SAME batch size: 1 untimed warm-up step, then up to `timed` full training steps
    (stopped early once `budget_s` seconds of timed work are spent, so the default bench run stays within minutes).
    For the default workload the reference's own CPU-runnable case C1 (BASELINE configs[0]: 256x128, bs 1) is timed
    beside it."""
    from oracle import ref_cpu
    from neurips18_hierchical_image_manipulation_amd import synth
    # torch/oneDNN on ALL 128 hardware threads of the GPU box is 8x slower than on 16-32 of them
    # (tools/cpu_thread_sweep.py: C1 0.8 s/step at 16-32 threads, 2.6 s at 64, 6-8 s at 128): time the baseline at its best
    cores = min(32, torch.get_num_threads())
    torch.set_num_threads(cores)
    bs, H, W = wl['bs'], wl['H'], wl['W']
    if name == 'box2mask':
        from oracle import ref_mask_cpu
        times = _time_oracle_steps(lambda: ref_mask_cpu.TwoStreamAEMask(),
                                   lambda s: synth.make_box2mask_batch(s, 0, bs, H, W), lambda m, b: m.step(b), timed, budget_s)
    else:
        times = _time_oracle_steps(lambda: ref_cpu.Mask2ImageModel(ref_cpu.Opt(**wl['flags'])),
                                   lambda s: synth.make_batch(s, 0, bs, H, W, wl['label_nc'], wl['color']),
                                   lambda m, b: m.optimize_parameters(b), timed, budget_s)
    sec = sum(times) / len(times)
    out = dict(value=round(bs / sec, 4), unit='images/s', cores=cores, cores_total=os.cpu_count(), kind='port',
               cpu=_cpu_model_name(),
               sample='%d timed full training step(s) after 1 warm-up step, %dx%d, batch %d (the bench workload itself), '
                      'torch CPU fp32 oracle, %d threads: %s s/step' % (len(times), W, H, bs, cores,
                                                                        '/'.join('%.1f' % t for t in times)))
    if name == 'c2':
        t1 = _time_oracle_steps(lambda: ref_cpu.Mask2ImageModel(ref_cpu.Opt(**C1)),
                                lambda s: synth.make_batch(s, 0, 1, 128, 256), lambda m, b: m.optimize_parameters(b), 3, 30.0)
        out['c1'] = dict(value=round(1.0 / (sum(t1) / len(t1)), 4), unit='images/s',
                         sample='BASELINE configs[0] (256x128, bs 1, 1-scale D): %d timed step(s) after 1 warm-up: %s s/step'
                                % (len(t1), '/'.join('%.2f' % t for t in t1)))
    return out


# --------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=3, help='timed oracle steps of the cpu_baseline leg')
    ap.add_argument('--fake-comm', action='store_true',
                    help='1-GPU stand-in for the data-parallel gradient exchange: after the normal timed run, a CHILD process '
                         'builds the same model with reducers attached BEFORE its first step (as a data-parallel rank has them) '
                         'whose buckets are device-to-device copies of the same bytes (730 MB G + 34 MB D at C2, 64 MB buckets) '
                         'on the comm streams at the real trigger points, times the same steps and reports the step-time '
                         'delta + the event-timed exposed waits ("fake_comm" in the JSON line)')
    ap.add_argument('--fake-comm-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--rccl-channels', type=int, default=0,
                    help='N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS; each channel is a workgroup = CUs taken '
                         'from the compute streams); 0 = the library default.  Printed under "ranks"')
    ap.add_argument('--tail-mb', type=float, default=8.0,
                    help='gradient reducer: size cap of the LAST bucket to become final (dist.GradReducer tail_bytes); 0 = no split')
    ap.add_argument('--variant', choices=['f4x4-resblock-fwd'], default=None,
                    help="OPT-IN reduced-work variant, reported as its OWN line (never the headline; VERDICT r4 item 7): "
                         "f4x4-resblock-fwd = the forward of the generator's ResnetBlock convolutions as Winograd F(4x4,3x3) "
                         "(HIM_ALGO_WINO4_TRAIN_FWD: 19.3 instead of 34.4 GFLOP per conv, ~3e-6 instead of 5e-7 relative "
                         "rounding per convolution); the line carries 'variant' and says so in 'dtype'")
    ap.add_argument('--g-backward-first', action='store_true',
                    help="A/B of the step order (DESIGN.md 6): loss_G.backward() BEFORE loss_D.backward() (the reference's own "
                         "order, train_mask2image.py:78-86) -- G's 730 MB exchange then has D's whole backward to hide under; "
                         "the shipped default runs loss_D.backward() first.  Printed under \"schedule\"")
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='c2',
                    help='c2 (default) = the BASELINE.json metric; c2local / c4 / box2mask = the other measured '
                         'configurations (DESIGN.md), same protocol and JSON schema')
    args = ap.parse_args()
    if args.rccl_channels > 0:
        os.environ['NCCL_MAX_NCHANNELS'] = str(args.rccl_channels)     # before the process group / the ranks exist
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args))

    from neurips18_hierchical_image_manipulation_amd import synth, config
    if args.g_backward_first:
        config.SCHED.d_backward_first = False
    if args.variant == 'f4x4-resblock-fwd':
        from neurips18_hierchical_image_manipulation_amd import ops as _ops
        from neurips18_hierchical_image_manipulation_amd._cabi import ALGO_WINO4_TRAIN_FWD
        _ops.current_algo().disable |= ALGO_WINO4_TRAIN_FWD
    from neurips18_hierchical_image_manipulation_amd.dist import (init_process_group_from_env, attach_data_parallel,
                                                                  replica_checksum_equal)
    from neurips18_hierchical_image_manipulation_amd.models import create_model

    rank, local, world = init_process_group_from_env()
    backend = check_world(args, world)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    wl = WORKLOADS[args.workload]
    bs, H, W = wl['bs'], wl['H'], wl['W']

    model = create_model(dict(wl['flags'], gpu_ids=[local], isTrain=True, checkpoints_dir='/tmp/him_bench',
                              name='bench_' + args.workload, batchSize=bs))
    # every rank draws the same Philox weights (and attach_data_parallel broadcasts rank 0's state anyway)
    model.netG.load_state_dict(synth.init_state_dict(model.netG.state_dict(), 1))
    model.netD.load_state_dict(synth.init_state_dict(model.netD.state_dict(), 2))
    attach_data_parallel(model, tail_bytes=int(args.tail_mb * (1 << 20)))
    if args.fake_comm_child:
        attach_data_parallel(model, fake=True, tail_bytes=int(args.tail_mb * (1 << 20)))

    # synthetic batches, resident in HBM before the timed region (4 distinct batches per rank, cycled; seeded by rank)
    if args.workload == 'box2mask':
        batches = [{k: v.to(device) for k, v in synth.make_box2mask_batch(s, rank, bs, H, W).items()} for s in range(4)]

        def step(i):
            b = batches[i % 4]
            names = ['G_Recon_comb', 'G_Recon_obj', 'KL_loss', 'loss_G_GAN', 'loss_D_GAN', 'loss_G_GAN_Feat']
            out = model.forward(b['label'], None, b['mask_ctx_in'], None, b['mask_out'], b['mask_obj_inst'], b['cls'],
                                b['mask_in'])[0]
            return dict(zip(names, out))
    else:
        batches = [{k: v.to(device) for k, v in synth.make_batch(s, rank, bs, H, W, wl['label_nc'], wl['color']).items()}
                   for s in range(4)]

        # resident before the timed region: the event tells the step that these device tensors are complete, so the input
        # encoding need not wait for whatever the previous step still has queued on the main stream (models/
        # pix2pixHD_condImg_model.py: optimize_parameters, 'ready_event')
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(device))
        for b in batches:
            b['ready_event'] = ready

        def step(i):
            return model.optimize_parameters(batches[i % 4])

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timing_on = (world > 1 or args.fake_comm_child) and hasattr(model, 'start_comm_timing')
    if timing_on:
        model.start_comm_timing()          # event pairs around every wait for the exchange (dist.timed_wait)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        losses = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    identical = exposed = fake = None
    if args.fake_comm_child:
        ex = model.read_comm_timing(args.steps)
        nbytes = sum(4 * r.flat.numel() for r in (model.reducer_G, model.reducer_D) if r is not None)
        print(json.dumps({'fake_comm_child': dict(ms_per_step=round(dt / args.steps * 1e3, 3), bytes_per_step=nbytes,
                                                  buckets=[len(r.buckets) for r in (model.reducer_G, model.reducer_D)
                                                           if r is not None], exposed_comm_ms=ex)}), flush=True)
        return
    if args.fake_comm and world == 1 and args.workload != 'box2mask':
        # The exchange stand-in (dist.GradReducer(fake=True)) in a CHILD process: same workload, same batches, reducers
        # attached before the first step.  (Rounds 3-4 attached them to THIS process's model after the timed run: the step
        # time of a model depends on what it ran before -- stream / allocator history; round 5 measured 56.4 ms for reducers
        # attached after a baseline phase with the early arena fill against 52.9 ms for the same reducers attached from the
        # start, profiles/r05_ab_log.txt -- and a data-parallel rank has them from the start.)
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(args.steps), '--warmup', str(max(args.warmup, 2)),
               '--workload', args.workload, '--tail-mb', str(args.tail_mb), '--fake-comm-child', '--no-roofline',
               '--no-cpu-baseline'] + (['--g-backward-first'] if args.g_backward_first else [])
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{"fake_comm_child"')]
        if r.returncode != 0 or not lines:
            raise SystemExit('bench.py --fake-comm: the child run failed:\n' + r.stdout[-1500:] + r.stderr[-3000:])
        ch = json.loads(lines[-1])['fake_comm_child']
        ex = ch['exposed_comm_ms']
        fake = dict(ms_per_step_without=round(dt / args.steps * 1e3, 3), ms_per_step_with=ch['ms_per_step'],
                    delta_ms=round(ch['ms_per_step'] - dt / args.steps * 1e3, 3), bytes_per_step=ch['bytes_per_step'],
                    buckets=ch['buckets'],
                    one_rank_of_n_will_post=dict(ms_per_step=ch['ms_per_step'],
                                                 images_per_s_per_gpu=round(WORKLOADS[args.workload]['bs'] * 1e3 / ch['ms_per_step'], 2),
                                                 note='the per-rank step of the data-parallel schedule (reducers attached from '
                                                      'the first step: no early arena fill, no early discriminator update, no '
                                                      'Adam split around the stem; Adam per bucket behind its exchange) with '
                                                      'the exchange as device-local copies -- xGMI time and the CU share of '
                                                      "RCCL's kernels come on top; what ONE rank of eight is expected to post"),
                    what='device-to-device copies of the gradient buckets on the optimizer streams at the real trigger '
                         'points (no second GPU), in a child process whose reducers are attached before its first step: '
                         'scheduling + HBM cost of the exchange, not xGMI time',
                    exposed_comm_ms=ex,
                    main_stream_upper_bound_ms=round(ex['g_update_tail'] + ex['d_update_wait'] +
                                                     min(ex.get('d_update_wait_real', 0.0),
                                                         ex.get('real_branch_join', 0.0)), 4))
    if world > 1:
        if timing_on:
            mine = model.read_comm_timing(args.steps)
            exposed = [None] * world
            dist.all_gather_object(exposed, mine)
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        identical = replica_checksum_equal(model)      # averaged gradients + identical start => identical replicas

    if rank == 0:
        ms = dt / args.steps * 1e3

        def _f(v):
            return round(float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else float(v), 5)
        out = {
            'metric': wl['metric'],
            'value': round(bs * world * args.steps / dt, 3), 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if not args.variant else 'f32 (Winograd F(4x4,3x3) in the ResnetBlock forward: 3e-6 instead of '
                                                    '5e-7 relative rounding per convolution; NOT the headline configuration)',
            'data': 'synthetic',
            'config': {'workload': wl['desc'], 'global_batch': bs * world, 'per_gpu_batch': bs,
                       'parallelism': 'dp%d' % world},
            'last_losses': {k: _f(v) for k, v in losses.items()},
            **({'variant': args.variant} if args.variant else {}),
            'schedule': {'d_backward_first': bool(config.SCHED.d_backward_first), 'adam_chunked': bool(config.SCHED.adam_chunked), 'adam_chunked_dp': bool(config.SCHED.adam_chunked_dp),
                         'd_from_ids': bool(config.SCHED.d_from_ids and config.SCHED.label_ids)},
        }
        if world > 1:
            out['ranks'] = {'world_size': dist.get_world_size(), 'backend': 'rccl' if backend == 'nccl' else backend,
                            'replicas_identical': identical}
            out['ranks']['rccl_max_nchannels'] = args.rccl_channels if args.rccl_channels > 0 else 'library default'
            if exposed is not None:
                # per rank, ms per step a stream sat idle for the gradient exchange (models/pix2pixHD_condImg_model.py
                # read_comm_timing): g_update_tail and d_update_wait are on the main stream, i.e. they delay the step
                out['exposed_comm_ms'] = {'per_rank': exposed,
                                          'main_stream_max_is': 'UPPER bound of what delays the step: g_update_tail + '
                                          'd_update_wait + min(d_update_wait_real, real_branch_join)',
                                          'main_stream_max': round(max(e['g_update_tail'] + e['d_update_wait'] +
                                                                       min(e.get('d_update_wait_real', 0.0),
                                                                           e.get('real_branch_join', 0.0))
                                                                       for e in exposed), 4)}
        if fake is not None:
            out['fake_comm'] = fake
        if not args.no_roofline:
            if args.workload == 'box2mask':
                out['roofline'] = direct_conv_roofline(device, bs, 256, 256, 3, 1, 1, 32, 32,
                                                       'direct fp32-MFMA conv3x3 256->256 @32x32 bs 32 (ResnetBlock latent convs '
                                                       'of MaskTwoStreamConvSwitch_NET)')
            else:
                ntiles = {'c2': 8 * 8 * 16, 'c2local': 8 * 4 * 8, 'c4': 16 * 8 * 8}[args.workload]
                out['roofline'] = dominant_kernel_roofline(device, bs, ntiles)
                out['g_forward'] = g_forward_roofline(model, batches[0], wl)
                if args.workload == 'c2':
                    out['step_roofline'] = step_flop_accounting(ms, bs, args.variant)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.workload, wl, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if identical is False:
        raise SystemExit('bench.py: data-parallel replicas diverged')


if __name__ == '__main__':
    main()
