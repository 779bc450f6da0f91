#!/bin/bash
# usage (GPU box): tools/ab_flrelu.sh <tag> <variant> [<variant> ...]   -- per-layer filtered_lrelu table (bf16, row-pitched, as the fused layer drives the
# kernels) for each variant library afcm_amd/csrc/variants/<variant>.so ("NEW" = the tree's own libafcm_hip.so), all on the same box; AB_FLRELU_FLAGS="--rotate 6": cold operands
tag=$1; shift
out=gpurun_out/${tag}_flrelu_ab.txt
: > $out
for v in "$@"; do
  if [ $v = NEW ]; then unset AFCM_HIP_LIB; else export AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/$v.so; fi
  echo "== $v" >> $out
  python tools/bench_flrelu.py --dtype bf16 --no-bias --raw pitched $AB_FLRELU_FLAGS >> $out 2>&1 || exit 1
done
grep -E "==|TOTAL" $out
