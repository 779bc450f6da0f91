"""``torch.ops.afcm.*``: the HIP kernels registered as PyTorch custom operators ("Plugin (FFI) surface").

The reference binds each CUDA plugin with pybind (``PYBIND11_MODULE`` in filtered_lrelu.cpp:294-298, upfirdn2d.cpp:102-105,
bias_act.cpp:94-97) and reaches it through ``custom_ops.get_plugin``.  The dispatcher-level equivalent is one operator library,
``afcm``, whose schemas carry the plugins' argument lists unchanged, plus the modulated convolution's forward / data-gradient
(same op, weights packed transposed) / weight-gradient triple that the reference leaves to cuDNN (NET:60-63):

    afcm::filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, writeSigns)
        -> (y, so, return_code)                                  filtered_lrelu.cpp:16-18
    afcm::filtered_lrelu_act_(x!, si, sx, sy, gain, slope, clamp, writeSigns) -> so       filtered_lrelu.cpp:213 (mutates x)
    afcm::upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> y      upfirdn2d.cpp:16
    afcm::bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> y          bias_act.cpp:32
    afcm::conv2d_pack_weights(w, dtype, mode) -> (packed, rows_pad)        mode 0: forward, 1: data gradient (transposed + flipped)
    afcm::conv2d(x, packed, oscale?, obias?, cout, ks, pad, rows_pad) -> y     y = oscale[n,o] * conv(w, x) + obias[o]
    afcm::conv2d_wgrad(dy, x, cout, cin, ks, pad) -> dw                        fp32 [cout, cin, ks, ks]

Kernels are registered for the ``CUDA`` dispatch key only (ROCm devices are ``device.type == 'cuda'``): a CPU tensor gets the
dispatcher's NotImplementedError -- there is no CPU implementation to fall back to.  Every implementation is the C ABI of
libafcm_hip.so (include/afcm_hip.h) behind the same Python launchers the op modules use; autograd stays where the
reference keeps it, in the ``torch.autograd.Function`` classes of ``torch_utils/ops/*.py``.
"""
import torch

from . import custom_ops as _co
from .ops import conv2d as _conv

_SCHEMAS = {
    'filtered_lrelu': '(Tensor x, Tensor fu, Tensor fd, Tensor b, Tensor si, int up, int down, int px0, int px1, int py0, int py1, '
                      'int sx, int sy, float gain, float slope, float clamp, bool flip_filter, bool writeSigns) -> (Tensor, Tensor, int)',
    'filtered_lrelu_act_': '(Tensor(a!) x, Tensor si, int sx, int sy, float gain, float slope, float clamp, bool writeSigns) -> Tensor',
    'upfirdn2d': '(Tensor x, Tensor f, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, bool flip, '
                 'float gain) -> Tensor',
    'bias_act': '(Tensor x, Tensor b, Tensor xref, Tensor yref, Tensor dy, int grad, int dim, int act, float alpha, float gain, '
                'float clamp) -> Tensor',
    'conv2d_pack_weights': '(Tensor w, ScalarType dtype, int mode) -> (Tensor, int)',
    'conv2d': '(Tensor x, Tensor packed, Tensor? oscale, Tensor? obias, int cout, int ks, int pad, int rows_pad) -> Tensor',
    'conv2d_wgrad': '(Tensor dy, Tensor x, int cout, int cin, int ks, int pad) -> Tensor',
}

_lib_def = torch.library.Library('afcm', 'DEF')
for _name, _schema in _SCHEMAS.items():
    _lib_def.define(_name + _schema)


def _filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filter, writeSigns):
    return _co._FilteredLReluPlugin.filtered_lrelu(x, fu, fd, b if b.numel() else None, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope,
                                                   clamp, flip_filter, writeSigns)


def _conv2d(x, packed, oscale, obias, cout, ks, pad, rows_pad):
    if x.ndim != 4 or packed.ndim != 4:
        raise RuntimeError(f'afcm::conv2d: expected a 4-D input and packed weights, got x{tuple(x.shape)} packed{tuple(packed.shape)}')
    return _conv._conv_raw(x.contiguous(), packed, int(rows_pad), oscale, int(cout), int(ks), int(pad), obias=obias)


def _conv2d_wgrad(dy, x, cout, cin, ks, pad):
    if dy.ndim != 4 or x.ndim != 4 or x.shape[1] != cin or dy.shape[1] != cout or dy.dtype != x.dtype:
        raise RuntimeError(f'afcm::conv2d_wgrad: incompatible operands dy{tuple(dy.shape)} x{tuple(x.shape)}')
    return _conv._wgrad_raw(dy.contiguous(), x.contiguous(), int(cout), int(cin), int(ks), int(pad))


_IMPLS = {
    'filtered_lrelu': _filtered_lrelu,
    'filtered_lrelu_act_': _co._FilteredLReluPlugin.filtered_lrelu_act_,
    'upfirdn2d': _co._Upfirdn2dPlugin.upfirdn2d,
    'bias_act': _co._BiasActPlugin.bias_act,
    'conv2d_pack_weights': lambda w, dtype, mode: _conv.pack_weights(w, dtype, int(mode)),
    'conv2d': _conv2d,
    'conv2d_wgrad': _conv2d_wgrad,
}
for _name, _fn in _IMPLS.items():
    _lib_def.impl(_name, _fn, 'CUDA')

OPS = tuple(_SCHEMAS)
