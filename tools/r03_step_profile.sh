#!/bin/bash
# usage: tools/r03_step_profile.sh <tag> [env assignments...]   (GPU box, repo root)
# kernel trace + FETCH_SIZE + WRITE_SIZE passes of the bench command (each in its own run), then the per-launch in-step table of
# the filtered_lrelu kernels (tools/flrelu_step_table.py) -> gpurun_out/<tag>_flrelu_step.txt
tag=$1; shift
for kv in "$@"; do export "$kv"; done
bash tools/pmc_traffic.sh ${tag}_traf bench.py --steps 3 --warmup 2 --lean --no-kernel-timing > gpurun_out/${tag}_traffic.log 2>&1
python tools/flrelu_step_table.py gpurun_out/${tag}_traf/trace --fetch gpurun_out/${tag}_traf/fetch --write gpurun_out/${tag}_traf/write > gpurun_out/${tag}_flrelu_step.txt 2>&1
python tools/traffic_summary.py gpurun_out/${tag}_traf gpurun_out/${tag}_pmc_traffic.json "${tag}" > gpurun_out/${tag}_bench_hbm_traffic.txt 2>&1
tail -4 gpurun_out/${tag}_flrelu_step.txt
