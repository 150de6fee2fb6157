#!/usr/bin/env python
"""One training step of a rocprofv3 kernel trace as a coarse text timeline: the step is cut into `nb` equal time bins and,
per bin and per stream, the kernel that occupied most of it is printed with the stream's busy fraction -- enough to see
which phase of the step (G forward, D / VGG passes, G backward, D backward, optimizer) the wall time goes to and which
stream is the critical one in it.  Usage: python tools/step_phases.py <db> [nbins] [step_index_from_end]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void\s+|him::|\(.*\)$', '', n)
    n = re.sub(r'at::native::.*', 'aten', n)
    return n[:34]


def main():
    db = sys.argv[1]
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end, queue_id, grid_x, grid_y, grid_z from kernels order by start').fetchall()
    from stepmarks import step_marks
    marks = step_marks(rows)        # a step ends with the generator's big Adam launch
    lo, hi = marks[-1 - back], marks[-back]
    rows = [r for r in rows if r[1] >= lo and r[2] <= hi + 1]
    span = hi - lo
    print('step window %.2f ms, %d launches' % (span / 1e6, len(rows)))
    qs = sorted(set(r[3] for r in rows))
    w = span / nb
    table = {q: [defaultdict(float) for _ in range(nb)] for q in qs}
    for n, s, e, q, gx, gy, gz in rows:
        b0, b1 = int((s - lo) / w), min(int((e - lo) / w), nb - 1)
        for b in range(max(b0, 0), b1 + 1):
            ov = min(e, lo + (b + 1) * w) - max(s, lo + b * w)
            if ov > 0:
                table[q][b][short(n)] += ov
    print('%7s ' % 'ms' + ' | '.join('%-42s' % ('queue %d' % q) for q in qs))
    for b in range(nb):
        cells = []
        for q in qs:
            d = table[q][b]
            if not d:
                cells.append('%-42s' % '')
                continue
            k = max(d, key=d.get)
            cells.append('%-34s %3d%%   ' % (k, 100 * sum(d.values()) / w))
        print('%7.2f ' % (b * w / 1e6) + ' | '.join(cells))


if __name__ == '__main__':
    main()
