#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (or a directory holding one) into a per-kernel table:
calls, total/avg/min/max duration, share of GPU time.  `--by-grid` splits a kernel by launch geometry
(= by layer shape for the conv kernels).  Used to produce the profiles/*.txt summaries."""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void\s+', '', name)
    name = name.replace('him::', '')
    name = re.sub(r'\(.*\)$', '', name)
    if len(name) > 70:
        name = name[:67] + '...'
    return name


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    by_grid = '--by-grid' in sys.argv
    path = args[0]
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*.db'), recursive=True))[0]
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute('pragma table_info(kernels)')]
    gcols = [x for x in ('grid_x', 'grid_y', 'grid_z', 'workgroup_x') if x in cols]
    q = 'select name, start, end%s from kernels' % (''.join(', ' + g for g in gcols))
    rows = c.execute(q).fetchall()
    stats, total = {}, 0
    for r in rows:
        name, s, e = r[0], r[1], r[2]
        key = short(name)
        if by_grid and gcols:
            g = r[3:]
            wg = g[3] if len(g) > 3 and g[3] else 1
            key += '  grid=(%d,%d,%d)' % (g[0] // wg, g[1], g[2])
        d = (e - s) / 1e3
        st = stats.setdefault(key, [0, 0.0, 1e30, 0.0])
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
        total += d
    print('%-95s %7s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
    for k, st in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print('%-95s %7d %12.1f %10.1f %10.1f %10.1f %6.2f' % (k, st[0], st[1], st[1] / st[0], st[2], st[3],
                                                             100 * st[1] / total))
    print('TOTAL kernel time %.1f us over %d dispatches' % (total, len(rows)))


if __name__ == '__main__':
    main()
