#!/bin/bash
# usage (GPU box): tools/ab_pitch_overhead.sh <value> [<value> ...]  -- the lean bench line with _rows.MAX_OVERHEAD set to each value (which planes are row-pitched)
for i in 1 2; do
for v in "$@"; do
  python - "$v" <<'PY' 2>/dev/null | tail -n 1 > /tmp/line.json || exit 1
import runpy, sys
import afcm_amd.torch_utils.ops._rows as r
r.MAX_OVERHEAD = float(sys.argv[1])
sys.argv = ['bench.py', '--lean', '--steps', '12', '--warmup', '3']
runpy.run_path('bench.py', run_name='__main__')
PY
  python - "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/line.json'))
k=d.get('kernels',{})
print(json.dumps({'max_overhead':sys.argv[1],'img_s':round(d['value'],1),'ms':round(d['ms_per_step'],2),**{n:round(v['ms_per_step'],2) for n,v in k.items()}}))
PY
done; done
