#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a built object (code-object metadata): tools/kernel_resources.py afcm_amd/csrc/filtered_lrelu_wave.o [filter]

Used by tests/test_abi.py to keep the hot filtered_lrelu kernels free of scratch spills ("check .vgpr_spill_count in
the code-object notes in CI, not by eye")."""
import os, re, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def kernel_resources(obj):
    """[{name (demangled), vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] of the gfx950 code object bundled in `obj`."""
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', tmp], cwd=d, capture_output=True, check=True)
        co = [f for f in os.listdir(d) if 'amdgcn' in f]
        if not co:
            raise RuntimeError(f'no device code object in {obj}')
        notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', os.path.join(d, co[0])], capture_output=True, text=True, check=True).stdout
    out, cur = [], None
    keymap = {'.vgpr_count': 'vgpr', '.agpr_count': 'agpr', '.sgpr_count': 'sgpr', '.vgpr_spill_count': 'vgpr_spill', '.sgpr_spill_count': 'sgpr_spill',
              '.private_segment_fixed_size': 'scratch', '.group_segment_fixed_size': 'lds', '.name': 'name'}
    for line in notes.splitlines():
        m = re.match(r'\s*(- )?(\.[a-z_]+):\s*(.*)$', line)
        if not m:
            continue
        k, v = m.group(2), m.group(3).strip()
        if m.group(1) and line.startswith('  - '):      # first key of a kernel record
            cur = {}
            out.append(cur)
        if cur is not None and k in keymap and (k != '.name' or 'name' not in cur or not line.startswith('      ')):
            if k == '.name':
                if not line.startswith('    .name'):
                    continue
                cur['name'] = v.strip("'")
            else:
                cur[keymap[k]] = int(v)
    out = [k for k in out if 'name' in k]
    names = '\n'.join(k['name'] for k in out)
    dem = subprocess.run(['c++filt'], input=names, capture_output=True, text=True).stdout.splitlines()
    for k, n in zip(out, dem):
        k['name'] = n
    return out


if __name__ == '__main__':
    filt = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in kernel_resources(sys.argv[1]):
        if filt in k['name']:
            n = re.sub(r'^void afcm::', '', k['name']).replace('__hip_bfloat16', 'bf16').replace('(afcm::FlreluMfmaParams)', '')
            print(f"{n[:70]:70s} vgpr {k.get('vgpr', 0):3d} agpr {k.get('agpr', 0):3d} sgpr {k.get('sgpr', 0):3d} spill {k.get('vgpr_spill', 0):3d} scratch {k.get('scratch', 0):4d} lds {k.get('lds', 0):6d}")
