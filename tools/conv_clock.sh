#!/bin/bash
# usage: tools/conv_clock.sh <tag> [conv_shape_probe args]  -- duration (kernel trace) and cycle counters (one --pmc pass, its own run) of the
# forward conv on one shape, with the LDS-patch kernel and with the gather kernel: effective shader clock = GRBM_GUI_ACTIVE / duration
tag=$1; shift
export TMPDIR=/tmp
for g in 0 1; do
  export AFCM_CONV_GATHER=$g
  d=gpurun_out/${tag}_g$g
  mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 tools/conv_shape_probe.py "$@" > $d/trace.log 2>&1 < /dev/null || true
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $d/pmc -- python3 tools/conv_shape_probe.py "$@" > $d/pmc.log 2>&1 < /dev/null || true
  echo "== gather=$g"
  grep -h "conv2d_fwd16" $d/trace/*/*kernel_stats.csv | cut -c1-200
  python3 - $d <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv2d_fwd16' not in k: continue
        acc[k[:40]][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[k[:40]] += 1
for k, v in acc.items():
    n = cnt[k]
    print(k, 'launches', n, ' '.join(f'{c}={x / n:.4g}' for c, x in sorted(v.items())))
PY
done
