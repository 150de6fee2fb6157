#!/usr/bin/env python
"""Where in the training step does NO matrix-pipe kernel run?  Walks the last whole step of a rocprofv3 kernel trace and
prints every interval without an MFMA kernel (merged when separated by less than `join_us`), in time order, with the
kernels that ran in it -- the phases in which the chip does only bandwidth / latency work.
Usage: python tools/nomfma_gaps.py <db> [min_us] [join_us] [step_index_from_end]"""
import re
import sqlite3
import sys
from collections import defaultdict

from timeline import MFMA


def short(n):
    n = re.sub(r'^void\s+|him::|\(.*\)$', '', n)
    n = re.sub(r'at::native::.*', 'aten', n)
    return n[:40]


def main():
    db = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    join_us = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
    back = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    c = sqlite3.connect(db)
    rows = c.execute('select name, start, end, queue_id from kernels order by start').fetchall()
    from stepmarks import step_marks
    marks = step_marks(rows)
    lo, hi = marks[-1 - back], marks[-back]
    rows = [r for r in rows if r[1] >= lo and r[2] <= hi + 1]
    mf = sorted((s, e) for n, s, e, q in rows if MFMA.search(n))
    # union of the MFMA intervals
    merged = []
    for s, e in mf:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    gaps = []
    last = lo
    for s, e in merged:
        if s > last:
            gaps.append([last, s])
        last = max(last, e)
    if hi > last:
        gaps.append([last, hi])
    # merge gaps separated by a short MFMA kernel
    j = []
    for g in gaps:
        if j and g[0] - j[-1][1] < join_us * 1e3:
            j[-1][1] = g[1]
            j[-1][2] += g[1] - g[0]
        else:
            j.append([g[0], g[1], g[1] - g[0]])
    tot = sum(g[1] - g[0] for g in gaps)
    print('step %.2f ms; no-MFMA time %.2f ms in %d intervals (%d after merging, listed when >= %.0f us)' % (
        (hi - lo) / 1e6, tot / 1e6, len(gaps), len(j), min_us))
    listed = 0
    for a, b, net in j:
        if net < min_us * 1e3:
            continue
        listed += net
        use = defaultdict(float)
        for n, s, e, q in rows:
            ov = min(e, b) - max(s, a)
            if ov > 0 and not MFMA.search(n):
                use['q%d %s' % (q, short(n))] += ov
        top = sorted(use.items(), key=lambda kv: -kv[1])[:5]
        print('%8.2f ms  +%7.1f us (no-MFMA %6.1f) : %s' % ((a - lo) / 1e6, (b - a) / 1e3, net / 1e3,
                                                             ', '.join('%s %.0f' % (k, v / 1e3) for k, v in top)))
    print('listed %.2f ms' % (listed / 1e6))


if __name__ == '__main__':
    main()
