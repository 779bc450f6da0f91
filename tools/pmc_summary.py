#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + PMC counters averaged per kernel name)."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + '/**/*kernel_stats.csv', recursive=True):
    print('==', f)
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 12:
            print(f"{row['Name'][:90]:90s} calls={row['Calls']:>5s} avg_ns={float(row['AverageNs']):12.0f} pct={row['Percentage']}")
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'][:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    print('==', f)
    for k, cs in acc.items():
        if 'flrelu' not in k and 'conv' not in k and 'mfma' not in k:
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print(f'   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})')
