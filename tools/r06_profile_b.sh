#!/bin/bash
# usage: tools/r06_profile_b.sh <tag>   (GPU box): kernel trace + HBM traffic passes of the bench command, per-kernel and per-layer tables
tag=$1
export TMPDIR=/tmp
bash tools/r03_step_profile.sh ${tag} < /dev/null > /dev/null 2>&1 || true
cp gpurun_out/${tag}_traf/trace/*/*kernel_stats.csv gpurun_out/${tag}_bench_kernel_stats.csv 2>/dev/null || true
python tools/step_kernel_table.py gpurun_out/${tag}_traf/trace --launches > gpurun_out/${tag}_step_kernels.txt 2>&1
head -8 gpurun_out/${tag}_step_kernels.txt; cat gpurun_out/${tag}_bench_hbm_traffic.txt; tail -4 gpurun_out/${tag}_flrelu_step.txt
python tools/bench_conv.py --dtype bf16 > gpurun_out/${tag}_conv_layers_bf16.txt 2>&1 < /dev/null || true
python tools/bench_flrelu.py --dtype bf16 --no-bias --raw pitched > gpurun_out/${tag}_flrelu_layers_bf16.txt 2>&1 < /dev/null || true
tail -1 gpurun_out/${tag}_conv_layers_bf16.txt; tail -2 gpurun_out/${tag}_flrelu_layers_bf16.txt
echo ALLDONE
