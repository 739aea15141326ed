#!/usr/bin/env python3
"""GPU idle time between kernels of a rocprofv3 kernel_trace.csv: where the device waits for the host.
usage: trace_gaps.py <kernel_trace.csv> [window_ms_from_end]

Kernels are sorted by start time; a gap is start[i+1] - max(end[0..i]).  Reported: busy / idle totals over the last `window` ms of the trace
(default: everything), a histogram of gap lengths, and the kernels most often found AFTER a gap > 20 us (the launch the host was late with)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline='') as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-50:]))
rows.sort()
if len(sys.argv) > 2:
    t_end = max(r[1] for r in rows)
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[2]) * 1e6]
span = max(r[1] for r in rows) - rows[0][0]
busy = 0
hist = defaultdict(lambda: [0, 0])
after = defaultdict(lambda: [0, 0])
pairs = defaultdict(lambda: [0, 0])
prev_name = None
cur_end = rows[0][0]
for s, e, name in rows:
    if s > cur_end:
        g = s - cur_end
        b = '<2us' if g < 2000 else '2-5us' if g < 5000 else '5-20us' if g < 20000 else '20-100us' if g < 100000 else '0.1-1ms' if g < 1000000 else '>1ms'
        hist[b][0] += 1
        hist[b][1] += g
        if g >= 20000:
            after[name][0] += 1
            after[name][1] += g
            pairs[(prev_name, name)][0] += 1
            pairs[(prev_name, name)][1] += g
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
    prev_name = name
print(f'kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %)  idle {(span - busy) / 1e6:.2f} ms')
for b in ('<2us', '2-5us', '5-20us', '20-100us', '0.1-1ms', '>1ms'):
    n, t = hist[b]
    print(f'  gaps {b:9s} n={n:6d}  total {t / 1e6:8.2f} ms ({100 * t / span:5.1f} % of span)')
print('kernels launched late (after a gap >= 20 us):')
for name, (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f'  {t / 1e6:8.2f} ms n={n:5d}  {name}')
print('gaps >= 20 us by (kernel before -> kernel after):')
for (a, b), (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'  {t / 1e6:8.2f} ms n={n:5d} avg {t / n / 1e3:7.1f} us  {str(a)[-38:]:38s} -> {b[-38:]}')
# timeline around the iteration boundaries: the kernels after the 3rd-last adam_multi_kernel (start of an iteration) and before the 2nd-last one (its end)
adam = [i for i, r in enumerate(rows) if 'adam_multi' in r[2]]
if len(adam) >= 3:
    def show(i0, i1, title):
        print(title)
        for i in range(max(i0, 1), min(i1, len(rows))):
            s, e, name = rows[i]
            print(f'  +{(s - rows[i0][0]) / 1e3:9.1f} us  gap {(s - max(r[1] for r in rows[max(0, i - 4):i])) / 1e3:7.1f}  dur {(e - s) / 1e3:8.1f}  {name[-60:]}')
    show(adam[-3], adam[-3] + 28, 'timeline: adam_multi_kernel of one iteration and the first kernels of the next')
    show(adam[-2] - 24, adam[-2] + 1, 'timeline: the last kernels of that iteration')
