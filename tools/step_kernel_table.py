#!/usr/bin/env python3
"""Every kernel of ONE bench.py step, by family, from a rocprofv3 kernel trace of that command.

    tools/step_kernel_table.py <dir with */*kernel_trace.csv> [--launches]

The LAST complete step (dispatches between the last two adam_multi_kernel launches): per kernel name launches, total us, average, min,
max, and the idle time between consecutive dispatches (end -> next start) summed over the step.  --launches: every launch of the kernels
outside the three hot families in order, with grid and duration (which plane dot / scale / copy is the slow one)."""
import argparse, collections, csv, glob, os, re, subprocess

_dm = {}


def demangle(name):
    if not name.startswith('_Z'):
        return name
    if name not in _dm:
        try:
            _dm[name] = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            _dm[name] = name
    return _dm[name]


def short(name):
    n = demangle(name)
    n = n.replace('__hip_bfloat16', 'bf16').replace('__bf16', 'bf16').replace('afcm::', '').replace('_Float16', 'f16').replace('void ', '')
    n = re.sub(r'\(.*', '', n)
    n = re.sub(r'at::native::', '', n)
    return n[:96]


def family(n):
    if 'conv2d_fwd' in n or 'conv2d_direct' in n:
        return 'conv2d'
    if 'conv2d_wgrad' in n:
        return 'conv2d_wgrad'
    if 'flrelu_wave' in n or 'flrelu_mfma_kernel' in n or 'flrelu_strip' in n or 'flrelu_sep' in n:
        return 'filtered_lrelu'
    return 'other'


ap = argparse.ArgumentParser()
ap.add_argument('trace')
ap.add_argument('--launches', action='store_true')
ap.add_argument('--spans', type=int, default=1, help='adam-to-adam intervals that make up one step (2 for the D + G iteration: the D update, then the G update)')
a = ap.parse_args()
rows = []
for f in glob.glob(os.path.join(a.trace, '**', '*kernel_trace.csv'), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_multi' in r['Kernel_Name']]
step = rows[idx[-1 - a.spans] + 1: idx[-1] + 1] if len(idx) > a.spans else rows
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step)
gaps = sum(max(0, int(b['Start_Timestamp']) - int(a_['End_Timestamp'])) for a_, b in zip(step, step[1:]))
print(f'# last step: {len(step)} dispatches, {(t1 - t0) / 1e6:.3f} ms first start -> last end, {busy / 1e6:.3f} ms of kernels, {gaps / 1e6:.3f} ms idle between dispatches')
fam = collections.defaultdict(lambda: [0, 0])
by = collections.defaultdict(list)
for r in step:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    n = short(r['Kernel_Name'])
    fam[family(n)][0] += 1
    fam[family(n)][1] += d
    by[n].append(d)
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f'# {k:16s} {c:4d} launches {t / 1e6:8.3f} ms')
print(f'{"launches":>8s} {"total us":>10s} {"avg":>8s} {"min":>8s} {"max":>8s}  kernel')
for n, ds in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'{len(ds):8d} {sum(ds) / 1e3:10.1f} {sum(ds) / len(ds) / 1e3:8.1f} {min(ds) / 1e3:8.1f} {max(ds) / 1e3:8.1f}  {n}')
if a.launches:
    print('# launches outside the three families, in order: us, grid, kernel')
    for r in step:
        n = short(r['Kernel_Name'])
        if family(n) == 'other':
            g = r.get('Grid_Size', r.get('Grid_Size_X', '?'))
            print(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} {g:>10s}  {n}')
