#!/bin/bash
# usage (GPU box): tools/ab_conv.sh <tag> <variant> [<variant> ...]  -- per-layer conv table (bf16, batch 16) for each variant library
# afcm_amd/csrc/variants/<variant>.so ("NEW" = the tree's own libafcm_hip.so), all on the same box; AB_CONV_FLAGS=--pitched: row-pitched operands
tag=$1; shift
out=gpurun_out/${tag}_conv_ab.txt
: > $out
for v in "$@"; do
  if [ $v = NEW ]; then unset AFCM_HIP_LIB; else export AFCM_HIP_LIB=$PWD/afcm_amd/csrc/variants/$v.so; fi
  echo "== $v" >> $out
  python tools/bench_conv.py --dtype bf16 --iters 10 $AB_CONV_FLAGS 2>/dev/null >> $out || exit 1
done
grep -E "==|TOTAL" $out
