#!/bin/bash
# round 5 evidence, part B (GPU box, repo root): the other configurations, per-layer tables, wave counters, 2-rank rehearsal
tag=r05
export TMPDIR=/tmp
bash tools/round_extra.sh ${tag} > gpurun_out/${tag}_bench_other_configs.txt 2>&1 < /dev/null
echo extra done; tail -9 gpurun_out/${tag}_bench_other_configs.txt
python tools/bench_conv.py --dtype bf16 > gpurun_out/${tag}_conv_layers_bf16.txt 2>&1 < /dev/null || true
python tools/bench_flrelu.py --dtype bf16 --no-bias --raw pitched > gpurun_out/${tag}_flrelu_layers_bf16.txt 2>&1 < /dev/null || true
python tools/bench_flrelu.py --dtype fp32 > gpurun_out/${tag}_flrelu_layers_fp32.txt 2>&1 < /dev/null || true
echo tables done
AFCM_BENCH_REHEARSE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 4 --lean > gpurun_out/${tag}_rehearse_gpus2.json 2> gpurun_out/${tag}_rehearse_gpus2.err < /dev/null; echo "rehearse rc=$?"
bash tools/pmc_wave.sh ${tag}_pmcw bench.py --steps 2 --warmup 1 --lean --no-kernel-timing < /dev/null > /dev/null 2>&1 || true
python tools/pmc_wave_table.py gpurun_out/${tag}_pmcw > gpurun_out/${tag}_flrelu_pmc_wave.txt 2>&1 || true
echo ALLDONE
