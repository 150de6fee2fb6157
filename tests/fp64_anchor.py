"""float64 anchor for the per-step (teacher-forced) parity tests  --  TEST INFRASTRUCTURE.

Two fp32 implementations of one training step differ from each other by their rounding; how much is "rounding" is a
per-tensor quantity (a 3-channel head's gradient sums 1e6 products, a 1024x1024x3x3 ResnetBlock filter's 4096).  The
yard-stick used here is the distance of the fp32 ORACLE (pinned bit-exactly to the reference, tests/golden/make_golden.py)
from the SAME step evaluated in float64:

    e32[q][name] = || q_fp32oracle[name] - q_fp64[name] ||_2 / || q_fp64[name] ||_2

for q in {grad, exp_avg, exp_avg_sq, delta (= parameter update of the step)}, per parameter tensor, per step.  It is
generated in the build container by ``python tests/fp64_anchor.py c1 c2`` -> tests/golden/fp64_anchor.json (numbers only)
and the GPU tests assert, per tensor, ``|| q_hip - q_fp64 || / || q_fp64 || <= K * max(e32, floor)`` with q_fp64 recomputed
on the GPU box's host by the same float64 oracle (the gradients themselves are 1.5 GB and do not travel).

The fp64 oracle is the oracle's own code (oracle/ref_cpu.py) constructed and run under
``torch.set_default_dtype(float64)``; before every step it adopts the fp32 oracle's parameters and Adam moments.
"""
import contextlib
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ANCHOR = os.path.join(HERE, 'golden', 'fp64_anchor.json')
QUANTITIES = ('grad', 'exp_avg', 'exp_avg_sq', 'delta')


@contextlib.contextmanager
def default_dtype(dt):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        yield
    finally:
        torch.set_default_dtype(prev)


def make_oracle(flags, dtype=torch.float32, device='cpu', yardstick=False):
    """oracle/ref_cpu.Mask2ImageModel with the build's seeded weights (G 1, D 2, VGG 3), in fp32 or fp64.
    ``device='cuda'`` (float64 only): the ANCHOR step runs through torch's own double-precision operators on the GPU --
    the same oracle code, the same float64 arithmetic, a summation order that differs from the host's at the 1e-15 level
    (tests/test_model_gpu.py::test_float64_anchor_on_the_gpu_equals_the_host_anchor).  The fp32 oracle -- the thing the
    HIP path is compared with -- always runs on the host.
    ``yardstick=True`` (round 6): the oracle's code in fp32 on torch's GPU operators (cuDNN/MIOpen switched off: ATen's native
    im2col + rocBLAS convolution, i.e. "the reference on another summation order", as tests/golden/chaos_envelope.json uses
    on the host) -- NOT a parity oracle, nothing is asserted against its values: it only counts how often an independent fp32
    implementation sits at a bimodal tensor's rounding baseline, over as many samples as the HIP path is sampled on (a host
    fp32 step costs 20 s at C2, this one a second)."""
    from oracle import ref_cpu
    from neurips18_hierchical_image_manipulation_amd import synth
    if device != 'cpu' and dtype != torch.float64 and not yardstick:
        raise ValueError('only the float64 anchor may leave the host: the fp32 oracle is pinned to the reference on the CPU')
    with default_dtype(dtype):
        om = ref_cpu.Mask2ImageModel(ref_cpu.Opt(**flags))
    om.anchor_device = torch.device(device)
    om.netG.load_state_dict(synth.init_state_dict(om.netG.state_dict(), 1))
    om.netD.load_state_dict(synth.init_state_dict(om.netD.state_dict(), 2))
    if om.vgg is not None:
        om.vgg.load_state_dict(synth.init_state_dict(om.vgg.state_dict(), 3, 'vgg'))
    if device != 'cpu':
        om.to(device)       # before the first optimizer step: the Adams hold the same Parameter objects
    return om


def adopt64(om64, om):
    """parameters + Adam state of the fp32 oracle -> the fp64 oracle (exact: every fp32 value is a double)."""
    om64.netG.load_state_dict(om.netG.state_dict())
    om64.netD.load_state_dict(om.netD.state_dict())
    for o64, o32 in ((om64.optimizer_G, om.optimizer_G), (om64.optimizer_D, om.optimizer_D)):
        if o32.state:
            o64.load_state_dict(o32.state_dict())      # torch casts the moments to the parameters' dtype (float64)
            for g64, g32 in zip(o64.param_groups, o32.param_groups):
                g64['lr'] = g32['lr']


def step64(om64, batch, **kw):
    dev = getattr(om64, 'anchor_device', torch.device('cpu'))
    # torch.device as a context: the oracle's factory calls (torch.zeros(...) for the one-hot / empty loss terms) land there
    with default_dtype(torch.float64), dev:
        b = type(batch)((k, (v.double() if torch.is_floating_point(v) else v).to(dev)) for k, v in batch.items())
        return om64.optimize_parameters(b, **kw)


def step32_yardstick(om32g, batch, **kw):
    """one fp32 step of a ``yardstick=True`` oracle (GPU, native convolutions)"""
    dev = om32g.anchor_device
    with torch.backends.cudnn.flags(enabled=False), dev:
        b = type(batch)((k, v.to(dev)) for k, v in batch.items())
        return om32g.optimize_parameters(b, **kw)


def snapshot(om):
    return {'G': {k: v.detach().clone() for k, v in om.netG.named_parameters()},
            'D': {k: v.detach().clone() for k, v in om.netD.named_parameters()}}


def rel_l2(a, b64):
    """|| a - b64 || / || b64 || in float64 -- on the GPU when one is there (MI355X adds doubles at half the fp32 rate; the
    190 M-element tensors of a full-size step take seconds per comparison on the host)."""
    dev = 'cuda' if torch.cuda.is_available() else 'cpu'
    a, b64 = a.detach().to(dev, torch.float64).reshape(-1), b64.detach().to(dev, torch.float64).reshape(-1)
    return float((a - b64).norm() / b64.norm().clamp_min(1e-300))


def oracle_quantities(om, before):
    """{'G/<name>': {grad, exp_avg, exp_avg_sq, delta}} of an oracle that has just taken its step from ``before``."""
    out = {}
    # the float64 subtraction of 183 M-element tensors takes seconds per step on the host: on the GPU when there is one
    dev = torch.device('cuda') if torch.cuda.is_available() else None
    for tag, net, opt in (('G', om.netG, om.optimizer_G), ('D', om.netD, om.optimizer_D)):
        for name, p in net.named_parameters():
            st = opt.state[p]
            d = dev or p.device
            out['%s/%s' % (tag, name)] = dict(grad=p.grad, exp_avg=st['exp_avg'], exp_avg_sq=st['exp_avg_sq'],
                                              delta=p.detach().to(d).double() - before[tag][name].to(d).double())
    return out


def dead_biases(net, tag):
    """'<tag>/<name>' of the conv biases that feed an InstanceNorm2d(affine=False) or a BatchNorm2d in training mode: their
    true gradient is exactly zero (the plane / batch mean is subtracted), every implementation holds rounding noise there --
    no relative error exists.  Sequential neighbours, and ``conv1`` -> ``norm1`` of the 'concat' FeatureFusionBlock."""
    names = set()
    convs, norms = ('Conv2d', 'ConvTranspose2d', 'SNConv2d'), ('InstanceNorm2d', 'BatchNorm2d')
    for mname, mod in net.named_modules():
        if mod.__class__.__name__ in ('Sequential', 'FusedSequential'):
            kids = list(mod.named_children())
            for (n0, c0), (_, c1) in zip(kids[:-1], kids[1:]):
                if c0.__class__.__name__ in convs and c1.__class__.__name__ in norms and getattr(c0, 'bias', None) is not None:
                    names.add('%s/%s%s.bias' % (tag, mname + '.' if mname else '', n0))
        elif mod.__class__.__name__ == 'FeatureFusionBlock' and hasattr(mod, 'conv1') and mod.conv1.bias is not None:
            names.add('%s/%s.conv1.bias' % (tag, mname))
    return names


def errors_vs_fp64(q, q64):
    """per-tensor relative L2 distance from the fp64 step: {name: {quantity: e}}."""
    return {name: {k: rel_l2(q[name][k], q64[name][k]) for k in QUANTITIES} for name in q64}


def load_anchor():
    with open(ANCHOR) as f:
        return json.load(f)


def generate(tag, steps):
    """build container: the fp32 oracle vs the fp64 oracle over ``steps`` teacher-forced steps of golden config ``tag``."""
    import time
    import numpy as np
    from neurips18_hierchical_image_manipulation_amd import synth
    g = np.load(os.path.join(HERE, 'golden', tag + '.npz'))
    flags = json.loads(str(g['flags']))
    B, H, W, color = int(g['B']), int(g['H']), int(g['W']), bool(int(g['color']))
    om, om64 = make_oracle(flags), make_oracle(flags, torch.float64)
    per_step = []
    for s in range(steps):
        t0 = time.time()
        adopt64(om64, om)
        before = snapshot(om)
        b = synth.make_batch(s, 0, B, H, W, flags.get('label_nc', 35), color)
        l32 = om.optimize_parameters(b)
        l64 = step64(om64, b)
        e = errors_vs_fp64(oracle_quantities(om, before), oracle_quantities(om64, before))
        for name in dead_biases(om.netG, 'G') | dead_biases(om.netD, 'D'):
            e.pop(name)
        lrel = max(abs(l32[k] - l64[k]) / max(abs(l64[k]), 1e-300) for k in l64)
        per_step.append(dict(loss_rel=lrel, tensors=e))
        worst = {k: max(v[k] for v in e.values()) for k in QUANTITIES}
        print('%s step %d: loss %.2e  worst grad %.2e  exp_avg %.2e  exp_avg_sq %.2e  delta %.2e  (%.0f s)' % (
            tag, s, lrel, worst['grad'], worst['exp_avg'], worst['exp_avg_sq'], worst['delta'], time.time() - t0),
            flush=True)
    return dict(flags=flags, B=B, H=H, W=W, threads=torch.get_num_threads(), steps=per_step)


if __name__ == '__main__':
    plan = {'c1': ('c1_traj', 20), 'c2': ('c2_traj', 5), 'tiny_global': ('tiny_global', 20), 'c4': ('c4_traj', 3)}
    res = load_anchor() if os.path.isfile(ANCHOR) else {}
    res['note'] = ('per step, per parameter tensor: ||q_fp32oracle - q_fp64|| / ||q_fp64|| for q in grad / exp_avg / '
                   'exp_avg_sq / delta (parameter update); teacher-forced along the fp32 oracle; tests/fp64_anchor.py')
    for key in (sys.argv[1:] or list(plan)):
        tag, steps = plan[key]
        res[key] = generate(tag, steps)
        with open(ANCHOR, 'w') as f:
            json.dump(res, f)
