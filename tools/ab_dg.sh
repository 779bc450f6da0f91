#!/bin/bash
# usage (GPU box): tools/ab_dg.sh <tag> <variant> [<variant> ...]  -- the full D + G iteration (bench.py --with-discriminator) for each variant
# library afcm_amd/csrc/variants/<variant>.so ("NEW" = the tree's own libafcm_hip.so), all on the same box: img/s and ms per iteration
tag=$1; shift
out=gpurun_out/${tag}_dg_ab.txt
: > $out
for v in "$@"; do
  if [ $v = NEW ]; then unset AFCM_HIP_LIB; else export AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/$v.so; fi
  timeout -k 10 300 python bench.py --lean --with-discriminator --steps 5 --warmup 2 2>/dev/null | tail -n 1 > /tmp/line.json || exit 1
  python - "$v" <<'PY' | tee -a $out
import json,sys
d=json.load(open('/tmp/line.json'))
print(f"{sys.argv[1]:12s} {d['value']:7.1f} img/s  {d['ms_per_step']:7.2f} ms per D + G iteration")
PY
done
