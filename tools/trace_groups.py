#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, workgroup) so per-shape costs are visible.
usage: trace_groups.py <kernel_trace.csv> [top]"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: [0, 0])
tot = 0
with open(sys.argv[1], newline='') as f:
    for r in csv.DictReader(f):
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        name = r['Kernel_Name'].split('(')[0][-60:]
        key = (name, '%sx%sx%s' % (r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']), r['Workgroup_Size_X'])
        rows[key][0] += 1
        rows[key][1] += d
        tot += d
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print(f'total kernel time {tot / 1e6:.2f} ms')
for k, (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{t / 1e6:9.3f} ms {100 * t / tot:5.2f}% n={n:5d} avg={t / n / 1e3:9.1f} us  {k[0]} grid={k[1]} wg={k[2]}')
