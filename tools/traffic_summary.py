#!/usr/bin/env python3
"""Per-kernel-family HBM traffic from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate runs).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of the bytes of wide coalesced
reads -> doubled; WRITE_SIZE is exact for streaming stores.  Units: the counters are in KiB-like 1024-byte units
(hbm_bytes = value * 1024)."""
import csv, glob, json, sys, collections

def load(d, name):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == name:
                k = row['Kernel_Name']
                fam = ('filtered_lrelu' if 'flrelu_mfma_kernel' in k or 'flrelu_sep' in k or 'flrelu_wave_kernel' in k else 'conv2d_wgrad' if ('wgrad' in k and 'reduce' not in k) else 'conv2d' if ('conv2d_fwd' in k or 'conv2d_direct' in k) else None)
                if fam:
                    acc[fam][0] += float(row['Counter_Value']); acc[fam][1] += 1
    return acc

root = sys.argv[1]
fetch, write = load(root + '/fetch', 'FETCH_SIZE'), load(root + '/write', 'WRITE_SIZE')
out = {}
for fam in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(fam, [0, 1]); w, nw = write.get(fam, [0, 1])
    rd = 2.0 * f * 1024 / max(nf, 1); wr = w * 1024 / max(nw, 1)
    out[fam] = dict(read_bytes_per_launch=rd, write_bytes_per_launch=wr, bytes_per_launch=rd + wr, launches_fetch_pass=nf, launches_write_pass=nw,
                    note='FETCH_SIZE doubled per the gfx950 correction; per-launch average over all launches of the family in a bench.py run')
    print(f'{fam:16s} read {rd/1e6:9.2f} MB  write {wr/1e6:9.2f} MB per launch  ({nf}/{nw} launches)')
if len(sys.argv) > 2:
    d = {k: v['bytes_per_launch'] for k, v in out.items()}
    # bench.py quotes this file as roofline.traffic: say which passes / build the bytes come from (argv[3], e.g. "r03, build <sha>")
    d['_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py' + (', ' + sys.argv[3] if len(sys.argv) > 3 else '')
    json.dump(d, open(sys.argv[2], 'w'), indent=1)
