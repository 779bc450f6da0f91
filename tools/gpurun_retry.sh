#!/bin/bash
# usage (build container): tools/gpurun_retry.sh <timeout> '<command>'  -- gpurun, again after a pause while it answers "no slot free" (exit 3:
# nothing ran, nothing was charged); any other exit code is final.
t=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 75
done
exit 3
