#!/usr/bin/env python3
"""Aggregate rocprofv3's marker (rocTX) trace by range name: calls, mean / total host-side duration, and the GPU kernel time that was
ENQUEUED inside each range (kernel-trace rows whose correlation falls between the range's push and pop on the same thread)."""
import csv, glob, sys
root = sys.argv[1]
mk = glob.glob(root + '/**/*marker_api_trace.csv', recursive=True)
if not mk:
    print('no marker trace found under', root); sys.exit(0)
rows = list(csv.DictReader(open(mk[0])))
agg = {}
for r in rows:
    name = r.get('Function') or r.get('Name') or '?'
    t0, t1 = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += t1 - t0
print(f'{"range":40s} {"calls":>7s} {"mean host ms":>13s} {"total host ms":>14s}')
for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{name[:40]:40s} {n:7d} {tot / n / 1e6:13.3f} {tot / 1e6:14.1f}')
