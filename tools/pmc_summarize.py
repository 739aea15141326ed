#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs: mean counter value per (kernel, counter), first dispatch dropped."""
import csv, sys, collections, glob
agg = collections.defaultdict(list)
for f in sys.argv[1:]:
    for path in glob.glob(f + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            k = r['Kernel_Name'].replace('void ', '').split('(')[0][:int(__import__('os').environ.get('PMC_NAME_LEN', '48'))]
            agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
for (k, c), v in sorted(agg.items()):
    vv = v[1:] if len(v) > 1 else v
    print(f'{k:50s} {c:28s} {sum(vv) / len(vv):18.1f} n={len(v)}')
