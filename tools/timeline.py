#!/usr/bin/env python
"""Timeline analysis of a rocprofv3 rocpd database: wall span, union-busy time, time with >=1 MFMA conv kernel
active, and the time during which ONLY non-MFMA kernels run (per kernel name) -- i.e. what sits on the critical
path next to the matrix pipe.  Usage: python tools/timeline.py <dir-or-db> [skip_fraction]"""
import glob
import os
import re
import sqlite3
import sys

MFMA = re.compile(r'gconv_fast_kernel|wgrad_fast_kernel|gconv_kernel|wgrad_kernel|wino_gemm_nt_kernel|wino_fused_kernel|wino_fused2_kernel|bgemm_kernel|bgemm_p_kernel|wgrad_fewch_mfma_kernel')


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    c = sqlite3.connect(path)
    rows = c.execute('select name, start, end from kernels order by start').fetchall()
    from stepmarks import step_marks
    adam = step_marks(rows)
    nsteps = 0
    if len(adam) >= 6:  # one mark per training step (the generator's big Adam launch): analyse the last 5 whole steps
        nsteps = 5
        lo, hi = adam[-6], adam[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    else:
        t0, t1 = rows[0][1], max(r[2] for r in rows)
        cut = t0 + (t1 - t0) * skip  # drop warm-up
        rows = [r for r in rows if r[1] >= cut]
    ev = []
    for i, (n, s, e) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    nm = 0
    last = ev[0][0]
    busy = mfma_t = 0
    alone = {}
    for t, kind, i in ev:
        dt = t - last
        if active:
            busy += dt
            if nm:
                mfma_t += dt
            else:
                for j in active:
                    k = re.sub(r'^void\s+|him::|\(.*\)$', '', rows[j][0])[:60]
                    alone[k] = alone.get(k, 0) + dt / len(active)
        last = t
        is_m = bool(MFMA.search(rows[i][0]))
        if kind:
            active.add(i)
            nm += is_m
        else:
            active.discard(i)
            nm -= is_m
    span = ev[-1][0] - ev[0][0]
    if nsteps:
        print('window = last %d training steps: %.2f ms/step' % (nsteps, span / 1e6 / nsteps))
    print('span %.1f ms  busy %.1f ms (%.1f%%)  mfma-active %.1f ms (%.1f%%)  idle %.1f ms' % (
        span / 1e6, busy / 1e6, 100 * busy / span, mfma_t / 1e6, 100 * mfma_t / span, (span - busy) / 1e6))
    print('time with NO MFMA kernel running, by kernel (%% of span):')
    for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:30]:
        print('  %-62s %8.2f ms %5.2f%%' % (k, v / 1e6, 100 * v / span))


if __name__ == '__main__':
    main()
