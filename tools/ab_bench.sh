#!/bin/bash
# usage (GPU box): tools/ab_bench.sh <tag> <variant> [<variant> ...]  -- the bench line (no CPU baseline) for each variant library
# afcm_amd/csrc/variants/<variant>.so ("NEW" = the tree's own libafcm_hip.so), all on the same box; prints img/s and the per-family ms
tag=$1; shift
out=gpurun_out/${tag}_bench_ab.jsonl
: > $out
for v in "$@"; do
  if [ $v = NEW ]; then unset AFCM_HIP_LIB; else export AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/$v.so; fi
  python bench.py --lean --steps 12 --warmup 3 2>/dev/null | tail -n 1 > /tmp/line.json || exit 1
  python - "$v" <<'PY' | tee -a $out
import json,sys
d=json.load(open('/tmp/line.json'))
k=d.get('kernels',{})
print(json.dumps({'variant':sys.argv[1],'img_s':round(d['value'],1),'ms':round(d['ms_per_step'],2),**{n:round(v['ms_per_step'],2) for n,v in k.items()}}))
PY
done
