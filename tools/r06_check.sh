#!/bin/bash
# usage: tools/r06_check.sh <tag>    (GPU box, repo root): GPU tests, a lean bench line + launch census, a kernel trace of the step with the per-kernel table
tag=$1
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1 < /dev/null; echo "tests rc=$?"; tail -3 gpurun_out/${tag}_tests.log
python bench.py --steps 20 --warmup 5 --cpu-baseline off --also off > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
print('img/s', round(d['value'], 1), 'ms', round(d['ms_per_step'], 2), 'host', round(d['host_ms_per_step'], 1), 'launches', d['launches_per_step'], 'outside ms', d['outside_the_three_families_ms'])
for k, v in d['kernels'].items(): print(' ', k, round(v['ms_per_step'], 2), round(v['frac'], 3))
PY
mkdir -p gpurun_out/${tag}_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 3 --warmup 2 --lean --no-kernel-timing > gpurun_out/${tag}_trace.log 2>&1 < /dev/null
python tools/step_kernel_table.py gpurun_out/${tag}_trace --launches > gpurun_out/${tag}_step_kernels.txt 2>&1
head -40 gpurun_out/${tag}_step_kernels.txt
