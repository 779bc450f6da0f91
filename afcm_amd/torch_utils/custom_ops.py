"""Drop-in for the reference's plugin loader (torch_utils/custom_ops.py:59, ``get_plugin(module_name, sources, headers,
source_dir, **build_kwargs)``).

The reference JIT-compiles a CUDA plugin per op and calls its pybind functions from the Python wrappers
(SG3OPS/filtered_lrelu.py:31,217; upfirdn2d.py:28,231; bias_act.py:47,143).  Here ``get_plugin`` compiles nothing: it returns
an object exposing the SAME functions with the SAME argument lists, each implemented on the C ABI of ``libafcm_hip.so``
(include/afcm_hip.h) -- so the reference's unmodified wrappers run on the HIP kernels when this module replaces theirs:

    filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, writeSigns)
        -> (y, so, return_code)                                   filtered_lrelu.cpp:16-18,296
    filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns) -> so      filtered_lrelu.cpp:213,297
    upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> y      upfirdn2d.cpp:16,104
    bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> y       bias_act.cpp:32,96

Error convention as in the reference: argument errors raise RuntimeError; "no fused kernel for this configuration" is not an
error but return_code = -1 with empty tensors (filtered_lrelu.cpp:52-56), on which the wrapper takes its generic path.
The sign tensor is opaque to the wrappers (they only hand ``so`` back as ``si`` with shifted sx / sy): its byte layout is this
library's, not the CUDA plugin's -- fp32 calls produce the reference's row-major packing as a 4-D tensor; 16-bit calls run the
matrix-core kernels, whose private layout is tagged by trailing unit dimensions (see filtered_lrelu below).
"""
import torch

from .. import _lib
from .ops import bias_act as _ba
from .ops import filtered_lrelu as _flr
from .ops import upfirdn2d as _ufd


def _none_if_empty(t):
    return None if (t is None or t.numel() == 0) else t


class _FilteredLReluPlugin:
    @staticmethod
    def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, writeSigns):
        si = _none_if_empty(si)
        if writeSigns and si is not None:
            raise RuntimeError('cannot read and write signs at the same time')        # filtered_lrelu.cpp:40
        empty = torch.empty([0], dtype=torch.uint8, device=x.device)
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            return torch.empty([0], dtype=x.dtype, device=x.device), empty, -1         # unsupported dtype: -1, as :218-219
        # identity filters arrive as 1x1 tensors (SG3OPS/filtered_lrelu.py:184-187): the C ABI takes NULL and a gain
        if fu.numel() == 1:
            gain, fu = gain * float(fu.reshape(-1)[0]), None
        if fd.numel() == 1:
            gain, fd = gain * float(fd.reshape(-1)[0]), None
        # 16-bit activations take the matrix-core kernels here as they do through afcm_amd.torch_utils.ops.filtered_lrelu.  The
        # plugin interface has no slot for the two things those need, so the shim carries them itself: the per-layer workspace of
        # constant fragments comes from the wrapper's cache (keyed by the filters' addresses and the configuration), and the sign
        # layout a WRITE call produced is tagged on the returned tensor as trailing unit dimensions (ndim - 4 = layout: 0 the
        # reference's row-major packing, 1 / 2 the matrix-core kernels' private ones).  The reference wrappers never look inside
        # ``so`` -- they save it and hand it back as ``si`` (SG3OPS/filtered_lrelu.py:222,258) -- so the tag survives the trip.
        layout = 0
        if si is not None:
            layout = si.ndim - 4
            if layout not in (0, 1, 2) or any(d != 1 for d in si.shape[4:]):
                raise RuntimeError('signs tensor was not produced by this plugin')
            si = si.reshape(si.shape[:4])
        cfg = (int(up), int(down), int(px0), int(px1), int(py0), int(py1), float(gain), float(slope), float(clamp), bool(flip_filter),
               int(sx), int(sy), layout)
        try:
            y, so, layout, _ = _flr._run(x, fu, fd, b, si, cfg, bool(writeSigns), allow_mfma=True, no_fallback=True)
        except _flr.NoFusedKernel:
            return torch.empty([0], dtype=x.dtype, device=x.device), empty, -1
        if so is not None and layout:
            so = so.reshape(list(so.shape) + [1] * layout)
        return y, (so if so is not None else empty), 0

    @staticmethod
    def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns):
        if si is not None and si.numel() and si.ndim != 4:
            raise RuntimeError('signs tensor was written by the matrix-core kernels: the unfused activation reads the row-major packing only')
        so = _flr._act_inplace(x, _none_if_empty(si), int(sx), int(sy), float(gain), float(slope), float(clamp), bool(writeSigns))
        return so if so is not None else torch.empty([0], dtype=torch.uint8, device=x.device)


class _Upfirdn2dPlugin:
    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        return _ufd._forward_raw(x, f, (int(upx), int(upy)), (int(downx), int(downy)), (int(padx0), int(padx1), int(pady0), int(pady1)),
                                 bool(flip), float(gain))


class _BiasActPlugin:
    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        spec = next(s for s in _ba.activation_funcs.values() if s.cuda_idx == int(act))
        return _ba._launch(x.contiguous(), _none_if_empty(b), _none_if_empty(xref), _none_if_empty(yref), _none_if_empty(dy), int(grad), int(dim),
                           spec, float(alpha), float(gain), float(clamp))


_PLUGINS = {'filtered_lrelu_plugin': _FilteredLReluPlugin, 'upfirdn2d_plugin': _Upfirdn2dPlugin, 'bias_act_plugin': _BiasActPlugin}


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):
    """Same signature as the reference's loader; sources / headers / build flags are ignored (nothing is compiled: the kernels
    live in libafcm_hip.so, built ahead of time by afcm_amd/csrc/Makefile).  Raises if the library is missing."""
    if module_name not in _PLUGINS:
        raise RuntimeError(f'no HIP implementation for plugin "{module_name}" (have: {sorted(_PLUGINS)})')
    _lib.load()
    return _PLUGINS[module_name]
