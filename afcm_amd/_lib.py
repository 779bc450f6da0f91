"""ctypes binding of libafcm_hip.so (C ABI declared in include/afcm_hip.h)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AFCM_HIP_LIB') or os.path.join(_HERE, 'libafcm_hip.so')     # AFCM_HIP_LIB: an alternative build (kernel experiments)

F32, F16, BF16 = 0, 1, 2
E_NOKERNEL, E_INVALID = -1, -2
SIGNS_NONE, SIGNS_WRITE, SIGNS_READ = 0, 1, 2

_DTYPES = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


class FilteredLReluArgs(C.Structure):
    """Mirror of `afcm_filtered_lrelu_args` (include/afcm_hip.h)."""
    _fields_ = [
        ('x', C.c_void_p), ('y', C.c_void_p), ('b', C.c_void_p), ('signs', C.c_void_p),
        ('fu', C.c_void_p), ('fd', C.c_void_p),
        ('dtype', C.c_int32),
        ('n', C.c_int32), ('c', C.c_int32), ('xh', C.c_int32), ('xw', C.c_int32), ('yh', C.c_int32), ('yw', C.c_int32),
        ('fuw', C.c_int32), ('fuh', C.c_int32), ('fdw', C.c_int32), ('fdh', C.c_int32),
        ('up', C.c_int32), ('down', C.c_int32),
        ('px0', C.c_int32), ('px1', C.c_int32), ('py0', C.c_int32), ('py1', C.c_int32),
        ('sx', C.c_int32), ('sy', C.c_int32), ('sh', C.c_int32), ('swb', C.c_int32),
        ('gain', C.c_float), ('slope', C.c_float), ('clamp', C.c_float),
        ('flip_filter', C.c_int32), ('sign_mode', C.c_int32),
        ('workspace', C.c_void_p), ('sign_layout', C.c_int32), ('plane_sum_slots', C.c_int32),
        ('plane_sum', C.c_void_p), ('oscale', C.c_void_p), ('skip', C.c_void_p), ('oscale2', C.c_void_p),
        ('x_pitch', C.c_int32), ('y_pitch', C.c_int32), ('skip_pitch', C.c_int32), ('row_pitch_ok', C.c_int32),
        ('clamp_flags', C.c_void_p),
    ]


AFFINE_MAX = 16


class AffineBank(C.Structure):
    """Mirror of `afcm_affine_bank` (include/afcm_hip.h)."""
    _fields_ = [
        ('layers', C.c_int32), ('n', C.c_int32), ('kw', C.c_int32), ('kg', C.c_int32),
        ('w_stride_n', C.c_int64), ('w_stride_l', C.c_int64),
        ('w', C.c_void_p), ('g', C.c_void_p),
        ('weight', C.c_void_p * AFFINE_MAX), ('bias', C.c_void_p * AFFINE_MAX),
        ('cout', C.c_int32 * AFFINE_MAX), ('w_index', C.c_int32 * AFFINE_MAX),
        ('alpha', C.c_float * AFFINE_MAX), ('beta', C.c_float * AFFINE_MAX),
    ]


MODULATION_MAX = 16


class ModulationLayer(C.Structure):
    """Mirror of `afcm_modulation_layer` (include/afcm_hip.h)."""
    _fields_ = [('cout', C.c_int32), ('cin', C.c_int32), ('kk', C.c_int32), ('demodulate', C.c_int32)] + [
        (k, C.c_void_p) for k in ('w', 't', 'magnitude', 'w_hat', 'wsq', 'scale', 's_eff', 'd', 'r', 'g_hat', 'g_s', 'g_d', 'dw', 'dt', 'workspace')]


PACK_MAX = 32


class PackEntry(C.Structure):
    """Mirror of `afcm_pack_entry` (include/afcm_hip.h)."""
    _fields_ = [('dst_fwd', C.c_void_p), ('dst_dgrad', C.c_void_p), ('w', C.c_void_p),
                ('cout', C.c_int32), ('cin', C.c_int32), ('rows_pad_fwd', C.c_int32), ('rows_pad_dgrad', C.c_int32)]


_lib = None
ABI_VERSION = 13

# name -> (restype, argtypes); every symbol include/afcm_hip.h declares must be listed here
# (tests/test_abi.py cross-checks this table against the header).
_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
SIGNATURES = {
    'afcm_abi_version': (C.c_int, []),
    'afcm_last_error': (C.c_char_p, []),
    'afcm_noop': (C.c_int, [_vp]),
    'afcm_filtered_lrelu_shapes': (C.c_int, [C.POINTER(FilteredLReluArgs)]),
    'afcm_filtered_lrelu': (C.c_int, [C.POINTER(FilteredLReluArgs), _vp]),
    'afcm_filtered_lrelu_workspace_bytes': (C.c_int64, []),
    'afcm_filtered_lrelu_prepare': (C.c_int, [C.POINTER(FilteredLReluArgs), _vp]),
    'afcm_filtered_lrelu_act': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _vp]),
    'afcm_upfirdn2d': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                 _i32, _i32, _i32, _f32, _vp]),
    'afcm_bias_act': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _f32, _f32, _f32, _vp]),
    'afcm_conv2d_block_k': (C.c_int, [_i32]),
    'afcm_conv2d_block_k_ks': (C.c_int, [_i32, _i32]),
    'afcm_conv2d_pack_weights_bk': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_pack_weights': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_pack_weights2': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_ld': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_wgrad_splits': (C.c_int, [_i32, _i32, _i32, _i32]),
    'afcm_conv2d_wgrad': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_wgrad_ld': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_wgrad_dots_ld': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_scale_planes': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    'afcm_plane_dot': (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    'afcm_plane_dot_ld': (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp]),
    'afcm_split16': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i64, _vp]),
    'afcm_conv2d_split': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.c_uint32, _i64, _vp, _vp, _vp]),
    'afcm_amax_bits': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    'afcm_unscale': (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'afcm_plane_dot_parts': (C.c_int, [_vp, _vp, _i64, _i32, _vp, _i32, _i64, _i32, _vp, _vp]),
    'afcm_conv2d_pack_split': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_uint32, _vp]),
    'afcm_plane_dot_gated_ld': (C.c_int, [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    'afcm_weight_norm_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'afcm_weight_norm_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'afcm_style_coefs_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'afcm_style_coefs_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    'afcm_layer_bwd_coefs': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    'afcm_affine_bank_workspace_bytes': (C.c_int64, [C.POINTER(AffineBank)]),
    'afcm_affine_bank_fwd': (C.c_int, [C.POINTER(AffineBank), C.POINTER(C.c_void_p), _vp]),
    'afcm_modulation_bank_workspace_floats': (C.c_int64, [_i32, _i32, _i32, _i32]),
    'afcm_modulation_bank_fwd': (C.c_int, [C.POINTER(ModulationLayer), _i32, _i32, _vp]),
    'afcm_modulation_bank_bwd': (C.c_int, [C.POINTER(ModulationLayer), _i32, _i32, _vp]),
    'afcm_conv2d_stride2': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    'afcm_conv2d_pack_bank': (C.c_int, [C.POINTER(PackEntry), _i32, _i32, _i32, _vp]),
    'afcm_affine_bank_bwd': (C.c_int, [C.POINTER(AffineBank), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _vp, _vp, _vp, _vp]),
    'afcm_pool_blocks_fwd': (C.c_int, [_vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    'afcm_pool_blocks_bwd': (C.c_int, [_vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    'afcm_l1_partials': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    'afcm_l1_grad': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _vp]),
    'afcm_axpy_planes': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _vp]),
    'afcm_fc_act_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    'afcm_fc_act_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp]),
    'afcm_mapping_input_fwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp]),
    'afcm_mapping_input_bwd_workspace_bytes': (C.c_int64, [_i32, _i32]),
    'afcm_mapping_input_bwd': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp]),
    'afcm_adam_chunk_elems': (C.c_int32, []),
    'afcm_adam_multi': (C.c_int, [_vp, _i32, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _i32, _vp]),
    'afcm_adam_multi_capturable': (C.c_int, [_vp, _i32, _i64, _vp, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _f32, _i32, _vp]),
}


def load():
    """Load the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP kernels are not built.  Run `python -c "import __graft_entry__ as g; g.build()"` '
                f'or `make -C afcm_amd/csrc`.  afcm_amd has no CPU fallback.')
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if lib.afcm_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libafcm_hip.so ABI version {lib.afcm_abi_version()} does not match this package ({ABI_VERSION}); rebuild it')
        _lib = lib
    return _lib


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f'afcm_amd kernels support float32/float16/bfloat16, got {t.dtype}') from None


def require_gpu(*tensors):
    """Every afcm_amd op runs on the GPU only: fail loudly otherwise."""
    for t in tensors:
        if t is not None and t.device.type != 'cuda':
            raise RuntimeError(
                f'afcm_amd ops need ROCm device tensors (got a tensor on {t.device}); there is no CPU path in this package '
                f'-- the aten reference lives in oracle/ and is test-only')


def stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def check(rc, what):
    """Translate a C-ABI status into the reference's error convention (RuntimeError <- TORCH_CHECK)."""
    if rc == 0 or rc == E_NOKERNEL:
        return rc
    if rc == E_INVALID:
        raise RuntimeError(f'{what}: {load().afcm_last_error().decode()}')
    raise RuntimeError(f'{what}: HIP error {rc - 1000} at kernel launch')
