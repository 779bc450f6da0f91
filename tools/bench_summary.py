#!/usr/bin/env python3
"""Pretty-print the JSON line of bench.py (stdin)."""
import json, sys
line = [l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
print(f"{d['value']:.1f} {d['unit']}  {d['ms_per_step']:.1f} ms/step  n_gpus={d['n_gpus']} dtype={d['dtype']}")
tot = 0
for k, v in d['kernels'].items():
    n = v.get('steps_timed') or d['steps']            # steps whose launches carried HIP events (bench.py --kernel-timing-every)
    per = v['total_ms'] / n
    tot += per
    print(f"  {k:16s} {per:8.1f} ms/step  {v['achieved']:8.1f} {v['unit']:8s} frac {v['frac']:.3f}  launches/step {v['launches'] / n:.0f}")
print(f"  other            {d['ms_per_step'] - tot:8.1f} ms/step")
if d.get('cpu_baseline'):
    print('  cpu_baseline', d['cpu_baseline'])
