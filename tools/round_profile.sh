#!/bin/bash
# usage: tools/round_profile.sh <tag>   (run on the GPU box, from the repo root)
# The round's evidence set: GPU tests, the bench line, smoke, a kernel trace of the bench command, HBM traffic and issue-slot
# counters (each --pmc pass in its own run, no trace domains next to --pmc), per-layer tables.  Everything lands in gpurun_out/.
set -e
tag=$1
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
echo tests done
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
echo bench done
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
bash tools/pmc_traffic.sh ${tag}_traf bench.py --steps 4 --warmup 2 --cpu-baseline off --no-kernel-timing > gpurun_out/${tag}_traffic.log 2>&1
python tools/traffic_summary.py gpurun_out/${tag}_traf gpurun_out/${tag}_pmc_traffic.json > gpurun_out/${tag}_bench_hbm_traffic.txt 2>&1
echo traffic done
bash tools/pmc_flrelu.sh ${tag}_pmc bench.py --steps 2 --warmup 1 --cpu-baseline off --no-kernel-timing > gpurun_out/${tag}_pmc.log 2>&1
python tools/pmc_table.py gpurun_out/${tag}_pmc > gpurun_out/${tag}_bench_pmc_raw.txt 2>&1
echo pmc done
python tools/bench_conv.py --dtype bf16 > gpurun_out/${tag}_conv_layers_bf16.txt 2>&1
python tools/bench_flrelu.py --dtype bf16 --no-bias > gpurun_out/${tag}_flrelu_layers_bf16.txt 2>&1
echo ALLDONE
