#!/usr/bin/env python3
"""Per-kernel table from tools/pmc_wave.sh output: counters per WAVE (counter / SQ_WAVES), quad-cycle counters x 4.
tools/pmc_wave_table.py <dir> [kernel-name substring, default flrelu_wave_kernel]"""
import csv, glob, re, sys, collections
d = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'flrelu_wave_kernel'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name']
        if want not in k:
            continue
        m = re.search(r'flrelu_wave_kernelI(DF16b|DF16_)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', k) or re.search(r'flrelu_wave_kernel<([^,]+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', k)
        key = ('<%s>' % ','.join(m.groups()[1:]) if m else k[:70]) + ' grid %s' % row['Grid_Size']
        acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
QUAD = ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
        'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC')
for key, cs in sorted(acc.items()):
    w = sum(cs['SQ_WAVES']) / max(1, len(cs['SQ_WAVES'])) if 'SQ_WAVES' in cs else 1.0
    g = lambda c: (sum(cs[c]) / len(cs[c]) / w * (4 if c in QUAD else 1)) if c in cs else float('nan')
    cyc = g('SQ_WAVE_CYCLES')
    print(f'{key}: waves {w:.0f}, cycles/wave {cyc:.0f}')
    print(f'   insts/wave: VALU {g("SQ_INSTS_VALU"):.0f}  MFMA {g("SQ_INSTS_MFMA"):.0f}  LDS {g("SQ_INSTS_LDS"):.0f}  SALU {g("SQ_INSTS_SALU"):.0f}  VMEM rd {g("SQ_INSTS_VMEM_RD"):.0f} wr {g("SQ_INSTS_VMEM_WR"):.0f}  SMEM {g("SQ_INSTS_SMEM"):.0f}')
    for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_MISC'):
        if c in cs:
            print(f'   {c:22s} {g(c):10.0f} cycles/wave = {100 * g(c) / cyc:5.1f} % of the wave\'s residency')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in cs:
        print(f'   MFMA busy {g("SQ_VALU_MFMA_BUSY_CYCLES"):.0f} cycles/wave = {100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / cyc:.1f} %;  LDS bank conflict cycles/wave {g("SQ_LDS_BANK_CONFLICT"):.0f}, LDS idx active {g("SQ_LDS_IDX_ACTIVE"):.0f}')
    for c in ('SQ_INST_LEVEL_VMEM', 'SQ_INST_LEVEL_LDS'):
        if c in cs:
            print(f'   {c} / wave-cycles = {sum(cs[c]) / len(cs[c]) / w / (cyc / 4):.2f} (average in flight per wave)')
