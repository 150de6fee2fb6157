#!/usr/bin/env python
"""Per-stream view of a rocprofv3 rocpd database: for the last N training steps (delimited by adam_kernel launches), the
busy time of every HIP stream / hardware queue, and for the busiest one (the main stream = the step's dependency chain)
its kernels by total time and the sum of the gaps between consecutive kernels.  Complements tools/timeline.py (which
merges all streams).  Usage: python tools/stream_view.py <dir-or-db> [nsteps]"""
import glob
import os
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'^void\s+|him::|\(.*\)$', '', n)
    return n if len(n) <= 64 else n[:61] + '...'


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    c = sqlite3.connect(path)
    rows = c.execute('select name, start, end, stream_id, queue_id from kernels order by start').fetchall()
    from stepmarks import step_marks
    adam = step_marks(rows)
    if len(adam) >= nsteps + 1:
        lo, hi = adam[-(nsteps + 1)], adam[-1]
        rows = [r for r in rows if r[1] >= lo and r[2] <= hi]
    else:
        nsteps = 1
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e6
    print('window: %d steps, %.2f ms/step, %d launches/step' % (nsteps, span / nsteps, len(rows) // nsteps))
    by = {}
    for r in rows:
        by.setdefault((r[3], r[4]), []).append(r)
    print('%-16s %8s %10s %10s' % ('stream/queue', 'launches', 'busy ms', 'ms/step'))
    for k, v in sorted(by.items(), key=lambda kv: -sum(x[2] - x[1] for x in kv[1])):
        busy = sum(x[2] - x[1] for x in v) / 1e6
        print('%-16s %8d %10.2f %10.2f' % ('%s/%s' % k, len(v) // nsteps, busy, busy / nsteps))
    main_k = max(by, key=lambda k: sum(x[2] - x[1] for x in by[k]))
    v = sorted(by[main_k], key=lambda x: x[1])
    gaps = sum(max(0, b[1] - a[2]) for a, b in zip(v[:-1], v[1:])) / 1e6
    big = sum(1 for a, b in zip(v[:-1], v[1:]) if b[1] - a[2] > 20000)
    print('main stream %s/%s: gaps between consecutive kernels %.2f ms/step (%d gaps > 20 us per step)' % (
        main_k + (gaps / nsteps, big // nsteps)))
    agg = {}
    for x in v:
        a = agg.setdefault(short(x[0]), [0, 0])
        a[0] += 1
        a[1] += x[2] - x[1]
    print('main-stream kernels by time (per step):')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print('  %-66s %5d %8.3f ms' % (k, n // nsteps, t / 1e6 / nsteps))


if __name__ == '__main__':
    main()
