#!/bin/bash
# usage: tools/fp32_prof.sh <tag>  -- kernel trace of the fp32 step (GPU box, repo root) -> gpurun_out/<tag>_fp32_kernel_stats.csv + a per-step table
tag=$1
export TMPDIR=/tmp
d=gpurun_out/${tag}_fp32_prof
rm -rf $d; mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --dtype fp32 --steps 4 --warmup 2 --lean --no-kernel-timing > $d/log.txt 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_fp32_kernel_stats.csv
rm -rf $d
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${tag}_fp32_kernel_stats.csv")))
steps = 6.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f"fp32 step (bench.py --dtype fp32, 2 warm-up + 4 timed steps traced): {tot / steps / 1e6:.1f} ms of kernels per step, {sum(int(r['Calls']) for r in rows) / steps:.0f} launches per step")
for r in rows[:24]:
    print(f"{int(r['TotalDurationNs']) / steps / 1e6:8.2f} ms/step {int(r['Calls']) / steps:7.1f} calls  {r['Name'][:130]}")
PY
