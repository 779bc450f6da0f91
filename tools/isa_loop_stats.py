#!/usr/bin/env python3
"""Instruction mix of a kernel's loops, from the built object's ISA:
    tools/isa_loop_stats.py afcm_amd/csrc/filtered_lrelu_wave.o flrelu_wave_kernelIDF16bLi2ELi2ELi64ELi32ELi1ELi1E   (mangled-name substring) [--dump]
Finds every backward branch (s_cbranch* to a lower address) and prints, for the loop body it closes, the number of instructions by class
(VALU / MFMA / SALU / LDS / VMEM / waitcnt ...), plus the mnemonic histogram of the largest loop.  Evidence tool: the counters say how
long a wave waits, the ISA says what for."""
import collections, os, re, subprocess, sys, tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, os.path.basename(obj))
        os.symlink(os.path.abspath(obj), tmp)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', tmp], cwd=d, capture_output=True, check=True)
        co = [f for f in os.listdir(d) if 'amdgcn' in f][0]
        return subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', os.path.join(d, co)], capture_output=True, text=True, check=True).stdout


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else names


def classify(m):
    if m.startswith('v_mfma') or m.startswith('v_smfmac'):
        return 'MFMA'
    if m.startswith('v_'):
        return 'VALU'
    if m.startswith('s_waitcnt'):
        return 'WAIT'
    if m.startswith('s_nop'):
        return 'NOP'
    if m.startswith('s_load') or m.startswith('s_buffer_load'):
        return 'SMEM'
    if m.startswith('s_'):
        return 'SALU'
    if m.startswith('ds_'):
        return 'LDS'
    if m.startswith('buffer_') or m.startswith('global_') or m.startswith('flat_') or m.startswith('scratch_'):
        return 'VMEM'
    return 'OTHER'


def kernels(text):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.*)>:$', line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):', line)
        if m and cur is not None:
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def main():
    obj, filt = sys.argv[1], sys.argv[2]
    dump = '--dump' in sys.argv
    ks = kernels(disassemble(obj))
    # objdump splits a kernel at every label: merge label sections back into the kernel they follow
    merged, cur = collections.OrderedDict(), None
    for k, ins in ks.items():
        if k.startswith('_Z') and '$local' not in k:
            cur = k
            merged.setdefault(cur, [])
        if cur is not None:
            merged[cur].extend(ins)
    for name, ins in merged.items():
        if filt not in name or not ins:
            continue
        addr = {a: i for i, (a, _, _) in enumerate(ins)}
        tot = collections.Counter(classify(m) for _, m, _ in ins)
        print(f'== {name}\n   whole kernel: {len(ins)} instructions  ' + '  '.join(f'{k} {v}' for k, v in sorted(tot.items())))
        loops = []
        for i, (a, m, ops) in enumerate(ins):
            if m.startswith('s_cbranch') or m == 's_branch':
                # relative branch: simm16 in dwords from the next instruction (objdump prints it unsigned)
                tgt = None
                mm = re.match(r'^(-?\d+)', ops)
                if mm:
                    imm = int(mm.group(1))
                    tgt = a + 4 + 4 * (imm - 65536 if imm >= 32768 else imm)
                if tgt is not None and tgt <= a and tgt in addr:
                    loops.append((addr[tgt], i))
        loops.sort(key=lambda l: l[0] - l[1])
        if '--inner' in sys.argv:
            # innermost loops only (no other loop strictly inside), largest first
            loops = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
        for (b, e) in loops[:6]:
            body = ins[b:e + 1]
            c = collections.Counter(classify(m) for _, m, _ in body)
            print(f'   loop [{ins[b][0]:#x}, {ins[e][0]:#x}]: {len(body)} instructions  ' + '  '.join(f'{k} {v}' for k, v in sorted(c.items())))
        if loops:
            b, e = loops[0]
            hist = collections.Counter(m for _, m, _ in ins[b:e + 1])
            print('   largest loop, mnemonics: ' + ', '.join(f'{m} {n}' for m, n in hist.most_common(60)))
            if dump:
                for a, m, ops in ins[b:e + 1]:
                    print(f'      {a:#x}  {m} {ops}')


if __name__ == '__main__':
    main()
