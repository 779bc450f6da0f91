"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a GPU,
and exports exactly the symbols include/afcm_hip.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'afcm_hip.h')


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(afcm_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def lib():
    from afcm_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/afcm_hip.h but not exported by libafcm_hip.so'


def test_binding_table_matches_header(lib):
    from afcm_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == _declared_symbols()


def test_abi_version(lib):
    assert lib.afcm_abi_version() == 13


def test_shapes_helper_is_pure_host(lib):
    """afcm_filtered_lrelu_shapes is host arithmetic (filtered_lrelu.cpp:61-94): enc0 of the 256^2 model."""
    from afcm_amd import _lib
    a = _lib.FilteredLReluArgs()
    a.n, a.c, a.xh, a.xw = 2, 3, 278, 278
    a.fuw, a.fuh, a.fdw, a.fdh = 12, 0, 12, 0
    a.up, a.down = 2, 2
    a.px0, a.px1, a.py0, a.py1 = 9, 8, 9, 8
    a.sign_mode = _lib.SIGNS_WRITE
    assert lib.afcm_filtered_lrelu_shapes(a) == 0
    assert (a.yh, a.yw) == (276, 276)
    assert (a.sh, a.swb) == (276 * 2 - 1 + 11, ((276 * 2 - 1 + 11 + 15) & ~15) // 4)
    a.px0 = -600                                         # upsampled buffer smaller than the down filter
    assert lib.afcm_filtered_lrelu_shapes(a) == _lib.E_INVALID
    assert b'upsampled buffer' in lib.afcm_last_error()


def test_struct_layout_matches_c():
    """sizeof/offsetof of the ctypes mirror vs the C compiler's view of the header."""
    from afcm_amd import _lib
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "afcm_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu\n", sizeof(afcm_filtered_lrelu_args), offsetof(afcm_filtered_lrelu_args, dtype),
  offsetof(afcm_filtered_lrelu_args, up), offsetof(afcm_filtered_lrelu_args, gain), offsetof(afcm_filtered_lrelu_args, sign_mode),
  offsetof(afcm_filtered_lrelu_args, clamp_flags)); return 0; }
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 't.c'), 'w').write(prog)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', os.path.join(d, 't'), os.path.join(d, 't.c')])
        out = subprocess.check_output([os.path.join(d, 't')]).decode().split()
    S = _lib.FilteredLReluArgs
    assert [int(v) for v in out] == [ctypes.sizeof(S), S.dtype.offset, S.up.offset, S.gain.offset, S.sign_mode.offset, S.clamp_flags.offset]


def test_ops_refuse_cpu_tensors():
    import torch
    from afcm_amd.torch_utils.ops import bias_act, filtered_lrelu, upfirdn2d
    x = torch.zeros(1, 1, 4, 4)
    for fn in (lambda: filtered_lrelu.filtered_lrelu(x), lambda: bias_act.bias_act(x, act='lrelu'),
               lambda: upfirdn2d.upfirdn2d(x, None)):
        with pytest.raises(RuntimeError, match='no CPU'):
            fn()


def test_operator_library_is_registered_with_the_plugin_schemas():
    """torch.ops.afcm.* exist after `import afcm_amd` (no native code needed to register), carry the plugins' argument lists
    (filtered_lrelu.cpp:16-18,213; upfirdn2d.cpp:16; bias_act.cpp:32) and have no CPU kernel."""
    import pytest
    import torch
    import afcm_amd  # noqa: F401
    from afcm_amd.torch_utils import op_registry
    for name in op_registry.OPS:
        assert hasattr(torch.ops.afcm, name), name
    s = str(torch.ops.afcm.filtered_lrelu.default._schema)
    assert 'Tensor si, int up, int down, int px0, int px1, int py0, int py1, int sx, int sy, float gain, float slope, float clamp, bool flip_filter, bool writeSigns' in s
    assert '-> (Tensor, Tensor, int)' in s
    assert 'Tensor(a!) x' in str(torch.ops.afcm.filtered_lrelu_act_.default._schema)      # the in-place mutation is declared
    assert 'int grad, int dim, int act, float alpha, float gain, float clamp' in str(torch.ops.afcm.bias_act.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.afcm.upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(1, 1), 1, 1, 1, 1, 0, 0, 0, 0, False, 1.0)


def _wave_kernel_table():
    """{(dtype, up, down, tow, toh, sign, epi): resources} of the built wave filtered_lrelu kernels (code-object metadata)."""
    import re
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    from kernel_resources import kernel_resources
    table = {}
    for obj, dt in (('filtered_lrelu_wave.o', 'bf16'), ('filtered_lrelu_wave_f16.o', 'f16')):
        path = os.path.join(root, 'afcm_amd', 'csrc', obj)
        if not os.path.exists(path):
            pytest.skip(f'{obj} not built (run __graft_entry__.build())')
        for k in kernel_resources(path):
            m = re.search(r'flrelu_wave_kernelI(?:DF16b|DF16_)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', k['name'])
            if not m:
                m = re.search(r'flrelu_wave_kernel<[^,]+, (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', k['name'])
            if m:
                table[(dt,) + tuple(int(v) for v in m.groups())] = k
    return table


def test_hot_filtered_lrelu_kernels_do_not_spill():
    """"zero scratch in <.,2,2,64,32,2,5> / <.,4,2,64,32,2,5> (check .vgpr_spill_count in the code-object notes in CI,
    not by eye)".  Every wave kernel the training step launches -- sign-writing forward with the plain / per-plane-factor / skip
    epilogues, aligned sign-reading transposed op (SIGN 3) with the plain / factor / factor + plane-sum epilogues -- must have NO
    scratch; the one-strip 48-row kernels of the 36^2 planes may keep a handful of dwords (measured faster at three waves per
    SIMD with <= 6 spilled registers than at two waves without)."""
    t = _wave_kernel_table()
    assert len(t) >= 100, len(t)
    WRITE, RA = 1, 3
    for dt in ('bf16', 'f16'):
        for (up, down, tow) in ((2, 2, 64), (2, 4, 32), (4, 2, 64)):
            for sign, epis in ((WRITE, (0, 1) + ((3,) if (up, down) == (2, 2) else ())), (RA, (0, 1, 5))):
                for epi in epis:
                    k = t[(dt, up, down, tow, 32, sign, epi)]
                    assert k.get('scratch', 0) == 0 and k.get('vgpr_spill', 0) == 0, (dt, up, down, sign, epi, k)
        for epi in (0, 1):
            k = t[(dt, 2, 2, 64, 48, WRITE, epi)]
            assert k.get('scratch', 0) == 0, k
        for epi in (0, 1, 5):
            k = t[(dt, 2, 2, 64, 48, RA, epi)]
            assert k.get('vgpr_spill', 0) <= 6, k
