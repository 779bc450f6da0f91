#!/bin/bash
# usage: tools/r06_profile_a.sh <tag>   (GPU box): the GPU test suite, the default bench line (all sub-records, CPU baseline), smoke
tag=$1
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -2 gpurun_out/${tag}_tests.log
s=$(date +%s)
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null; echo "bench rc=$? in $(( $(date +%s) - s )) s"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 < /dev/null; echo "smoke rc=$?"; tail -2 gpurun_out/${tag}_smoke.log
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
print('img/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'host', round(d['host_ms_per_step'], 1), 'outside', d['launches_per_step'].get('outside_the_three_families'), round(d['outside_the_three_families_ms'], 2))
for k, v in d['kernels'].items(): print(' ', k, round(v['ms_per_step'], 2), round(v['frac'], 3))
for k, v in d['also'].items(): print(' also', k, {a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a in ('images_per_sec', 'ms_per_step', 'host_ms_per_replay', 'error')})
print(' cpu', d['cpu_baseline']['value'])
PY
