#!/usr/bin/env python
"""profiles/r<NN>_dominant_kernel_rocprof.json: the dominant launch (batched Winograd GEMM; round 4: bgemm_kernel<0,1> -- and
<1,1> for the data gradients on the forward panel, <0,0> for the weight gradients on the kept input transform -- on a grid
of 16 * (1024/128)^2 = 1024 workgroups; rounds 2-3: gconv_fast_kernel on 2048 / 1024) as rocprofv3 --kernel-trace saw it
(a) alone, in the trace of tools/gemm_bench.py, and (b) inside the traced training step of bench.py.
Usage: python tools/dominant_kernel_json.py <gemm_trace dir|db> <step trace dir|db> <out.json> [tag of the step trace]"""
import glob
import json
import os
import sqlite3
import sys

PEAK, FLOP = 157.3, 2.0 * 16 * 1024 ** 3
KERNELS, GRID = ('bgemm_kernel<0, 1>', 'bgemm_kernel<1, 1>', 'bgemm_kernel<0, 0>'), 1024
KERNEL = 'bgemm_kernel<0,1> / <1,1> / <0,0>'


def durations(path, need_adam):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    c = sqlite3.connect(path)
    rows = c.execute('select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start').fetchall()
    if need_adam:       # whole training steps only: between the first and the last Adam launch, warm-up steps dropped
        from stepmarks import step_marks
        adam = step_marks(rows)
        lo, hi = adam[len(adam) // 3], adam[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    hit = [r for r in rows if any(k in r[0] for k in KERNELS) and r[3] // max(r[6], 1) == GRID and r[4] == 1 and r[5] == 1]
    sel = [r[2] - r[1] for r in hit]
    # UNION of the launches' intervals: the data- and weight-gradient GEMMs of a layer run CONCURRENTLY on two streams (each
    # then takes twice as long, two progress at once); union time / launches = the wall-clock the chip spends per GEMM
    union, end = 0, -1
    for r in sorted(hit, key=lambda r: r[1]):
        lo, hi = max(r[1], end), r[2]
        if hi > lo:
            union += hi - lo
        end = max(end, r[2])
    durations.union_ns = union
    durations.by_kernel = {}
    for r in hit:
        durations.by_kernel.setdefault(r[0].split('(')[0].strip(), []).append(r[2] - r[1])
    return sel


def main():
    iso, step = durations(sys.argv[1], False), durations(sys.argv[2], True)
    iso = iso[3:] if len(iso) > 6 else iso        # warm-up launches of the microbench
    out = dict(N=1024, kernel='%s grid %d' % (KERNEL, GRID), launches=len(iso),
               avg_launch_ms=round(sum(iso) / len(iso) / 1e6, 4),
               executed_tflops=round(FLOP / (sum(iso) / len(iso) * 1e-9) / 1e12, 2))
    out['frac_of_f32_mfma_peak'] = round(out['executed_tflops'] / PEAK, 4)
    if step:
        avg = sum(step) / len(step)
        out.update(in_step_launches=len(step), in_step_avg_launch_ms=round(avg / 1e6, 4),
                   in_step_executed_tflops=round(FLOP / (avg * 1e-9) / 1e12, 2))
        out['in_step_frac_of_f32_mfma_peak'] = round(out['in_step_executed_tflops'] / PEAK, 4)
        out['in_step_avg_launch_ms_by_kernel'] = {k: round(sum(v) / len(v) / 1e6, 4) for k, v in durations.by_kernel.items()}
        # wall-clock per GEMM = union of the (partly concurrent) launch intervals / launches
        un = durations.union_ns / len(step)
        out.update(in_step_union_ms_per_launch=round(un / 1e6, 4),
                   in_step_union_frac_of_f32_mfma_peak=round(FLOP / (un * 1e-9) / 1e12 / PEAK, 4))
    if len(sys.argv) > 4:
        out['in_step_source'] = ('profiles/%s_* (tools/trace_unbound.sh: a trace that is not launch-starved, every launch of 8 '
                                 'steps queued behind a spin kernel)' % sys.argv[4])
    out['command'] = ('rocprofv3 --kernel-trace --stats -- python tools/gemm_bench.py 20 (isolated) and rocprofv3 --kernel-trace '
                      '-- python bench.py --steps 6 --warmup 3 (in step); tools/collect_profiles.sh')
    with open(sys.argv[3], 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
