#!/bin/bash
# usage (GPU box): tools/pmc_conv_layers.sh <tag>  -- wave counters (tools/pmc_wave.sh: three --pmc passes, no trace domains) of the conv kernels on
# four layer shapes of the generator + a kernel trace of the same command for the wall time: matrix-pipe occupancy and in-kernel clock per kernel
tag=$1
export TMPDIR=/tmp
out=gpurun_out/${tag}_conv_pmc_layers.txt
: > $out
for shape in "64 64 276" "362 512 148" "512 512 84" "512 512 36"; do
  set -- $shape
  d=${tag}_pmc_$1_$2_$3
  bash tools/pmc_wave.sh $d tools/prof_conv.py --cin $1 --cout $2 --hw $3 > /dev/null 2>&1
  mkdir -p gpurun_out/$d/trace
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$d/trace -- python3 tools/prof_conv.py --cin $1 --cout $2 --hw $3 > gpurun_out/$d/trace.log 2>&1 || true
  echo "==== $1 -> $2 channels at $3^2, batch 16, bf16" >> $out
  python tools/pmc_wave_table.py gpurun_out/$d conv2d_ >> $out 2>&1
  python - >> $out 2>&1 <<PY
import csv, glob
for f in glob.glob('gpurun_out/$d/trace/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv2d_fwd16x' in r['Name'] or 'wgrad16g' in r['Name']:
            print('   wall: %-60s calls %s avg %.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
grep -E "====|cycles/wave|MFMA busy|WAIT_ANY|WAIT_INST_ANY |wall:" $out
