#!/bin/bash
# usage (GPU box): tools/ab_zero_stuff.sh -- the D + G iteration with the stride-2 backward's zero-stuffed dy made by upfirdn2d (up 2, one tap: one
# pass) against the fill + strided copy it replaces (module switch afcm_amd.torch_utils.ops.conv2d.ZERO_STUFF_UPFIRDN)
for v in False True False True; do
python - "$v" <<'PY' 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value'],1), 'img/s', round(d['ms_per_step'],2), 'ms')" "ZERO_STUFF_UPFIRDN=$v"
import sys, runpy
import afcm_amd.torch_utils.ops.conv2d as C
C.ZERO_STUFF_UPFIRDN = sys.argv[1] == 'True'
sys.argv = ['bench.py', '--lean', '--with-discriminator', '--steps', '5', '--warmup', '2']
runpy.run_path('bench.py', run_name='__main__')
PY
done
