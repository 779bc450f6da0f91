#!/bin/bash
# usage: tools/r06_dg.sh <tag>   (GPU box): kernel trace of the full D + G iteration, per-kernel table of one iteration (two adam-to-adam spans)
tag=$1
export TMPDIR=/tmp
mkdir -p gpurun_out/${tag}_dgtrace
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_dgtrace -- python3 bench.py --with-discriminator --steps 3 --warmup 2 --lean --no-kernel-timing > gpurun_out/${tag}_dgtrace.log 2>&1 < /dev/null
python tools/step_kernel_table.py gpurun_out/${tag}_dgtrace --spans 2 > gpurun_out/${tag}_dg_step_kernels.txt 2>&1
head -45 gpurun_out/${tag}_dg_step_kernels.txt
