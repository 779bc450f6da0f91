#!/usr/bin/env python3
"""Per-kernel averages of the counters of the three --pmc passes tools/pmc_wave.sh writes: tools/pmc_kernel_table.py <dir> <name filter>"""
import collections, csv, glob, sys
d, flt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(d + '/pmc*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if flt not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k in sorted(acc):
    print(k[:150])
    v = {c: x / cnt[k][c] for c, x in acc[k].items()}
    for c in sorted(v): print(f'    {c:28s} {v[c]:14.4g}')
    wc = v.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA'):
            if c in v: print(f'    {c + " / WAVE_CYCLES":40s} {v[c] / wc:8.3f}')
    if 'SQ_LDS_IDX_ACTIVE' in v and 'SQ_LDS_BANK_CONFLICT' in v: print(f'    LDS conflict share {v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]:.3f}')
    if 'SQ_BUSY_CYCLES' in v and 'SQ_LDS_IDX_ACTIVE' in v: print(f'    LDS active / busy cycles {v["SQ_LDS_IDX_ACTIVE"] / v["SQ_BUSY_CYCLES"]:.3f}')
