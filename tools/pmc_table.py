#!/usr/bin/env python3
import csv, glob, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ''
for f in sorted(glob.glob(d + '/**/*kernel_stats.csv', recursive=True)):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8: print(f"{row['Name'][:80]:80s} calls={row['Calls']:>5s} avg_us={float(row['AverageNs'])/1e3:10.1f} pct={row['Percentage']}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        if pat in row['Kernel_Name']:
            acc[row['Kernel_Name'][:80]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in acc.items():
    print(k)
    w = sum(cs.get('SQ_WAVES', [1])) / max(1, len(cs.get('SQ_WAVES', [1])))
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print(f'   {c:28s} {m:16.0f}   per-wave {m / w:12.1f}')
