#!/usr/bin/env python3
"""Overlap of collective kernels with compute kernels in a rocprofv3 --kernel-trace run (kernel_trace.csv).
usage: overlap_summary.py <trace dir>   -- prints, per collective launch, the compute kernels running concurrently."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def fam(n):
    if 'nccl' in n.lower() or 'rccl' in n.lower(): return 'rccl'
    for k, v in (('conv2d_wgrad', 'conv2d_wgrad'), ('conv2d_fwd', 'conv2d'), ('flrelu', 'filtered_lrelu'), ('adam', 'adam')):
        if k in n: return v
    return None
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), fam(r['Kernel_Name']), r['Kernel_Name'][:60], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
coll = [e for e in ev if e[2] == 'rccl']
comp = [e for e in ev if e[2] in ('conv2d', 'conv2d_wgrad', 'filtered_lrelu')]
print(f'{len(rows)} kernel launches, {len(coll)} collective launches, {len(comp)} conv / filtered_lrelu launches')
tot = ov = 0
per = collections.Counter()
for s, e, _, name, q in coll:
    o = 0
    for cs, ce, cf, cn, cq in comp:
        lo, hi = max(s, cs), min(e, ce)
        if hi > lo:
            o += hi - lo; per[cf] += hi - lo
    tot += e - s; ov += min(o, e - s)
if coll:
    print(f'collective time {tot/1e6:.3f} ms, of which {ov/1e6:.3f} ms ({100*ov/max(tot,1):.0f} %) ran while a conv / filtered_lrelu kernel was executing')
    print('overlapped with:', {k: round(v / 1e6, 3) for k, v in per.items()}, 'ms')
    streams = collections.Counter(q for *_, q in coll)
    print('collective queues/streams:', dict(streams), '| compute:', dict(collections.Counter(q for *_, q in comp).most_common(3)))
