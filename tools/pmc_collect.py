#!/usr/bin/env python
"""Fold the rocprofv3 --pmc csv outputs (one counter group per pass) for one kernel into the JSON kept under
profiles/.  Usage: python tools/pmc_collect.py <kernel-substring> <grid_x_workgroups> out.json dir1 dir2 ...
FETCH_SIZE / WRITE_SIZE are KB per launch; gfx950 FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md,
HBM section) -> corrected fetch = 2 x raw."""
import csv
import glob
import json
import os
import sys


def main():
    kern, out = sys.argv[1], sys.argv[3]
    want_grid = int(sys.argv[2])
    vals = {}
    for d in sys.argv[4:]:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    if kern not in row.get('Kernel_Name', ''):
                        continue
                    gx = int(row.get('Grid_Size_X', row.get('Grid_Size', 0)) or 0)
                    wg = int(row.get('Workgroup_Size_X', row.get('Workgroup_Size', 256)) or 256)
                    # some rocprofv3 builds report Grid_Size = x * y * z threads: (1128, 4, 1) x 256 -> 4512 workgroups
                    nwg = gx // max(wg, 1)
                    if want_grid and nwg != want_grid and gx != want_grid * wg and not (nwg % want_grid == 0 and
                                                                                        nwg // want_grid <= 64):
                        continue
                    vals.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    res = {k: sum(v) / len(v) for k, v in vals.items()}
    res['launches_averaged'] = {k: len(v) for k, v in vals.items()}
    if 'FETCH_SIZE' in res:
        res['fetch_bytes_raw'] = res['FETCH_SIZE'] * 1024
        res['fetch_bytes_corrected'] = 2 * res['fetch_bytes_raw']
    if 'WRITE_SIZE' in res:
        res['write_bytes'] = res['WRITE_SIZE'] * 1024
    if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
        res['traffic_bytes_corrected'] = res['fetch_bytes_corrected'] + res['write_bytes']
    if 'TCC_HIT_sum' in res and 'TCC_MISS_sum' in res:
        res['l2_hit_rate'] = res['TCC_HIT_sum'] / (res['TCC_HIT_sum'] + res['TCC_MISS_sum'])
    with open(out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
