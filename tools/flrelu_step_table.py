#!/usr/bin/env python3
"""Per-launch table of the filtered_lrelu kernels INSIDE a bench.py step, from rocprofv3 output of that command.

    tools/flrelu_step_table.py <dir with */*kernel_trace.csv> [--fetch <dir>] [--write <dir>] [--batch 16]

Every flrelu_wave / flrelu_mfma / flrelu_sep dispatch of the LAST timed step (dispatches are split into steps at the
adam_multi_kernel launches) in launch order: template arguments, grid, duration, and -- the launches of one step visit the
generator's layers in a fixed order (forward: enc0..13, L0..L14; backward: reversed) -- the layer, its algorithmic bytes
 and GB/s.  With --fetch / --write (separate --pmc passes of the same command) also the HBM bytes of each
launch (FETCH_SIZE doubled per the gfx950 correction, 1024-byte units) and their ratio to the algorithmic bytes.
VERDICT r02 'next' 1c: explains the gap between the isolated per-layer table (tools/bench_flrelu.py) and the in-step time."""
import argparse
import re, csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, '**', pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def is_fl(name):
    return 'flrelu_wave_kernel' in name or 'flrelu_mfma_kernel' in name or 'flrelu_sep' in name


_dm = {}


def demangle(name):
    if not name.startswith('_Z'):
        return name
    if name not in _dm:
        import subprocess
        for tool in ('/opt/rocm/lib/llvm/bin/llvm-cxxfilt', 'c++filt'):
            try:
                _dm[name] = subprocess.run([tool, name], capture_output=True, text=True).stdout.strip() or name
                break
            except OSError:
                _dm[name] = name
    return _dm[name]


def short(name):
    name = demangle(name)
    m = re.search(r'(flrelu_\w+?)_kernel<([^>]*)>', name)
    if not m:
        return name[:40]
    args = m.group(2).replace('__hip_bfloat16', 'bf16').replace('__bf16', 'bf16').replace('afcm::', '').replace('_Float16', 'f16').replace(' ', '')
    return m.group(1).replace('flrelu_', '') + '<' + args + '>'


def last_step(disp):
    """dispatches (sorted by start) of the last complete step: between the last two adam launches"""
    idx = [i for i, r in enumerate(disp) if 'adam_multi' in r['Kernel_Name']]
    if len(idx) < 2:
        return disp
    return disp[idx[-2] + 1: idx[-1] + 1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--res', type=int, default=256)
    a = ap.parse_args()
    from afcm_amd import layer_schedule as sched
    pl = sched.plan(a.res, 4, 1, {})
    layers = [L for L in pl['enc'] + pl['dec']]

    def alg_bytes(L):
        h = L['in_size'] + L['k'] - 1
        o = L['out_size']
        if L['up'] == 1 and L['down'] == 1:
            return None
        sh = o * L['down'] - (L['down'] - 1) + len(L['fd']) - 1
        sw4 = (sh + 15) // 16 * 4
        return a.batch * L['cout'] * ((h * h + o * o) * 2 + sh * sw4)

    disp = sorted(rows(a.trace, '*kernel_trace.csv'), key=lambda r: int(r['Start_Timestamp']))
    step = last_step(disp)
    t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
    fl = [r for r in step if is_fl(r['Kernel_Name'])]
    pmc = {}
    for key, d, cname in (('rd', a.fetch, 'FETCH_SIZE'), ('wr', a.write, 'WRITE_SIZE')):
        if not d:
            continue
        rr = sorted([r for r in rows(d, '*counter_collection.csv') if r['Counter_Name'] == cname], key=lambda r: int(r['Dispatch_Id']))
        st = last_step(rr)
        pmc[key] = [float(r['Counter_Value']) * 1024 * (2.0 if key == 'rd' else 1.0) for r in st if is_fl(r['Kernel_Name'])]
    # layer of each launch: resampling layers only (the 1x1 ToRGB layer runs bias_act, not these kernels)
    res_layers = [L for L in layers if not (L['up'] == 1 and L['down'] == 1)]
    order = res_layers + res_layers[::-1]
    tot = tot_b = 0.0
    print(f'step: {(t1 - t0) / 1e6:.2f} ms kernel span, {len(fl)} filtered_lrelu launches')
    print(f'{"#":>3s} {"layer":14s} {"dir":3s} {"kernel":30s} {"grid":>7s} {"VGPR":>4s} {"scr":>4s} {"us":>8s} {"alg MB":>8s} {"GB/s":>7s}' + ('  HBM rd MB  wr MB  ratio' if pmc else ''))
    fam = {}
    for i, r in enumerate(fl):
        L = order[i] if len(fl) == len(order) else None
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        ab = alg_bytes(L) if L else None
        # a forward launch of a decoder layer with an encoder skip (EPI bit 2, the kernel's last template argument) also reads the skip
        # tensor -- bytes the fused op needs, though SURVEY 8(d)'s per-op figure (x + y + signs) does not count them: ratio2 includes them
        m = re.search(r'Li(\d+)EEEv', r['Kernel_Name'])
        skip_b = a.batch * L['cout'] * L['out_size'] ** 2 * 2 if (L and m and int(m.group(1)) & 2) else 0
        line = f'{i:3d} {(L["name"] if L else "?"):14s} {("fwd" if i < len(res_layers) else "bwd"):3s} {short(r["Kernel_Name"]):30s} {int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]):7d} {r["VGPR_Count"]:>4s} {r["Scratch_Size"]:>4s} {us:8.1f}'
        if ab:
            line += f' {ab / 1e6:8.1f} {ab / us / 1e3:7.0f}'
            tot_b += ab
        if pmc and all(len(v) == len(fl) for v in pmc.values()):
            rd, wr = pmc.get('rd', [0] * len(fl))[i], pmc.get('wr', [0] * len(fl))[i]
            line += f'  {rd / 1e6:9.1f} {wr / 1e6:6.1f}' + (f'  {(rd + wr) / ab:5.2f}' if ab else '') + (f'  ({(rd + wr) / (ab + skip_b):4.2f} with the skip operand)' if (ab and skip_b) else '')
            k = ('fwd' if i < len(res_layers) else 'bwd')
            f = fam.setdefault(k, [0.0, 0.0, 0.0])
            f[0] += rd; f[1] += wr; f[2] += ab or 0
        tot += us
        print(line)
    print(f'filtered_lrelu in the step: {tot / 1e3:.2f} ms, {tot_b / 1e6:.0f} MB algorithmic = {tot_b / tot / 1e3:.0f} GB/s')
    for k, (rd, wr, ab) in fam.items():
        print(f'  {k}: HBM read {rd / 1e6:.0f} MB + write {wr / 1e6:.0f} MB = {(rd + wr) / max(ab, 1):.2f} x algorithmic ({ab / 1e6:.0f} MB)')


if __name__ == '__main__':
    main()
