#!/usr/bin/env python
"""profiles/r<NN>_roofline_table.txt: per layer of the C2 step, every MFMA kernel it launches -- shape, executed FLOP,
isolated duration (the layer alone under rocprofv3, tools/layer_bench.py), average duration of the SAME (kernel, grid) inside
the traced training step, TFLOP/s and fraction of the fp32 MFMA peak for both.

    python tools/roofline_table.py <dir with <layer>.txt by-grid summaries> <step by-grid summary> <traced steps>
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from layer_bench import LAYERS      # noqa: E402

PEAK = 157.3
MFMA = ('gconv_fast_kernel', 'wgrad_fast_kernel', 'bgemm_kernel', 'bgemm_p_kernel', 'wino_fused_kernel', 'wino_fused2_kernel', 'wgrad_kernel', 'gconv_kernel',
        'wgrad_fewch_mfma_kernel')
ROW = re.compile(r'^(.*?)\s{2}grid=\((\d+),(\d+),(\d+)\)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)')


def parse(path):
    out = {}
    with open(path) as f:
        for line in f:
            m = ROW.match(line)
            if m:
                out[(m.group(1).strip(), tuple(int(m.group(i)) for i in (2, 3, 4)))] = dict(
                    calls=int(m.group(5)), total=float(m.group(6)), avg=float(m.group(7)), mn=float(m.group(8)))
    return out


def executed_gflop(kernel, spec):
    """FLOP one launch of ``kernel`` EXECUTES for this layer (direct form 2*MAC; Winograd F(2x2): /2.25, F(4x4): /4 with the
    tile padding ignored)."""
    kind, B, Cin, H, W, Cout, k, s, p, mode, frozen, _ = spec
    if kind == 'conv':
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    else:
        OH, OW = H * s, W * s
    direct = 2.0 * B * OH * OW * Cin * Cout * k * k / (s * s if kind == 'deconv' else 1) / 1e9
    if kernel.startswith('bgemm_'):
        return direct / (4.0 if (frozen and min(Cin, Cout) >= 128) else 2.25)      # HimAlgo.wino4_min_c = 128
    if kernel.startswith('wino_fused'):
        return direct / 2.25
    return direct


def main():
    layer_dir, step_path, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    step = parse(step_path)
    print('fp32 MFMA peak %.1f TFLOP/s.  isolated = the layer alone (tools/layer_bench.py under rocprofv3 --kernel-trace, avg of 6 '
          'launches); in-step = the same (kernel, grid) in the traced training step (all streams busy).  One launch of a kernel '
          'may serve several layers of the same shape (calls/step): the in-step average is then over ALL of them (g_down4 forward and '
          'g_up1 data gradient are the same launch shape; d0_l3 and d1_l3 share the weight-gradient grid (32,4,6): read their in-step '
          'TF/s as a range, the isolated column is per layer; the PERSISTENT kernels -- wino_fused2_kernel on one workgroup per compute unit, bgemm_p_kernel on two -- launch the same grid for every layer, so their in-step columns average over all of VGG conv1_2 + conv2_1, resp. conv2_2 .. conv3_4).  PMC passes of the rows marked in DESIGN.md: profiles/r0N_pmc_*.json.' % PEAK)
    print('%-9s %-36s %-42s %-14s %8s %9s %7s %6s %9s %7s %6s %6s' % (
        'layer', 'what', 'kernel', 'grid', 'GFLOP', 'isol_us', 'TF/s', 'frac', 'instep_us', 'TF/s', 'frac', 'calls'))
    tot = {}
    for name, spec in LAYERS.items():
        path = os.path.join(layer_dir, name + '.txt')
        if not os.path.isfile(path):
            continue
        for (kern, grid), st in sorted(parse(path).items(), key=lambda kv: -kv[1]['total']):
            if not kern.startswith(MFMA):
                continue
            gf = executed_gflop(kern, spec)
            iso = st['avg']
            ins = step.get((kern, grid))
            isf = gf / iso * 1e3
            line = '%-9s %-36s %-42s %-14s %8.2f %9.1f %7.1f %6.3f' % (name, spec[-1][:36], kern[:42], str(grid).replace(' ', ''),
                                                                     gf, iso, isf, isf / PEAK)
            if ins:
                inf = gf / ins["avg"] * 1e3
                line += ' %9.1f %7.1f %6.3f %6.1f' % (ins['avg'], inf, inf / PEAK, ins['calls'] / steps)
                tot[(kern, grid)] = ins['total'] / steps
            else:
                line += ' %9s %7s %6s %6s' % ('-', '-', '-', '-')
            print(line)
    print('in-step kernel time covered by the rows above: %.2f ms per step' % (sum(tot.values()) / 1e3))


if __name__ == '__main__':
    main()
