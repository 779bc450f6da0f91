#!/bin/bash
# usage: tools/pmc_traffic.sh <outdir> <python script + args...>
# HBM traffic of a command: kernel trace + FETCH_SIZE and WRITE_SIZE, each in its own --pmc pass (the two do not fit one
# pass on gfx950: TCC counter budget, MI355X_MICROARCH.md).  The program itself follows `--` (no shell hop).
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/trace -- python3 "$@" > gpurun_out/$out/trace.log 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$out/fetch -- python3 "$@" > gpurun_out/$out/fetch.log 2>&1 || true
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$out/write -- python3 "$@" > gpurun_out/$out/write.log 2>&1 || true
tail -2 gpurun_out/$out/fetch.log gpurun_out/$out/write.log | cut -c1-200
