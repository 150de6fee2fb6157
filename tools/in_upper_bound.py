#!/usr/bin/env python
"""TIMING EXPERIMENT, results are numerically meaningless: the ceiling of what fusing conv -> InstanceNorm -> ReLU for the
generator's stride-2 convolutions and transposed convolutions ('n1') could buy in the C2 step.

The fused design would (1) accumulate the plane statistics in the producing convolution's epilogue and (2) normalise + ReLU in
the NEXT convolution's loader, so that the stand-alone `instnorm_fwd` launch, its read of y and its write of z disappear
while y is still written once and read once (+ once more by the InstanceNorm backward).  This script runs bench.py's own
timed loop with exactly that traffic and NO loader cost at all: `ops._InstNorm.forward` returns its input for every
InstanceNorm whose plane is a power of two >= 512 pixels (the generator's stem, four down-convolutions and four transposed
convolutions; the discriminator's planes are odd-sized and the ResnetBlocks have their own fused unit) with mean 0 / rstd 1,
launching nothing.  The backward is untouched (the fused design keeps it).  The step-time difference to the normal line is
an UPPER bound of n1's gain: a real fusion adds per-element work to loaders that are on the critical path and cannot use
the range-checked zero padding (a padded zero would become relu(-mean * rstd)).

    python tools/in_upper_bound.py [bench.py flags]     -> one bench JSON line tagged "experiment": "in_upper_bound"
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


def main():
    import torch
    from neurips18_hierchical_image_manipulation_amd import ops
    const, skipped = {}, [0]
    normal = ops._InstNorm.forward

    def forward(ctx, x, residual, act, slope, eps):
        B, Cn, H, W = x.shape
        hw = H * W
        if residual is not None or hw < 512 or hw & (hw - 1):
            return normal(ctx, x, residual, act, slope, eps)
        ctx.set_materialize_grads(False)
        x = x.contiguous()
        key = (B * Cn, x.device)
        if key not in const:
            const[key] = (torch.zeros(B * Cn, device=x.device), torch.ones(B * Cn, device=x.device))
        ctx.x, (ctx.mean, ctx.rstd) = x, const[key]
        ctx.cfg = (B * Cn, hw, act, slope)
        ctx.has_res = False
        skipped[0] += 1
        return x.view_as(x)

    ops._InstNorm.forward = staticmethod(forward)
    import bench
    import io
    import contextlib
    buf = io.StringIO()
    sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline'] + sys.argv[1:]
    with contextlib.redirect_stdout(buf):
        bench.main()
    line = json.loads(buf.getvalue().strip().splitlines()[-1])
    line['experiment'] = 'in_upper_bound'
    line['instnorm_fwd_launches_skipped_total'] = skipped[0]
    line['note'] = ('timing only: the generator stem / down / up InstanceNorm forwards are skipped (no launch, no z), outputs are '
                    'NOT the model\'s; compare ms_per_step with the normal line of the same box')
    print(json.dumps(line))


if __name__ == '__main__':
    main()
