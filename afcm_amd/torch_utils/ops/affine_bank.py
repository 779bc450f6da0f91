"""The style affine layers of all SynthesisLayers as ONE op (C ABI afcm_affine_bank_*, csrc/affine_bank.hip).

Every SynthesisLayer begins with ``styles = self.affine(torch.cat((w, global_w), 1))`` (NET:349-352), an equalised-lr
FullyConnectedLayer (NET:69-104) from the 512 + 1024 latent / global features to the layer's input channels; ToRGB multiplies the
result by 1 / sqrt(Cin k^2) (NET:351).  ``affine_bank(ws, global_w, specs)`` returns the same styles for a list of layers from one
launch, and its backward produces every weight / bias gradient and the gradients of ``ws`` and ``global_w`` from three: the framework
path is 15 cat + 15 GEMM launches forward and 30 GEMMs + 15 bias reductions + 14 accumulations backward, 4-10 us each.
First-order only (``once_differentiable``): the generator's step never differentiates the styles twice; a caller that does uses the
layers one by one (``SynthesisLayer.modulation``).  ``supported()`` says whether the kernels take the shapes."""
import ctypes as C

import torch

from ... import _lib


class Spec:
    """One layer of the bank: the FC module (weight [cout, kw + kg], bias [cout], its gains), the index of its latent in ws and an
    extra factor on the styles (ToRGB)."""

    def __init__(self, fc, w_index, scale=1.0):
        self.fc, self.w_index, self.scale = fc, int(w_index), float(scale)


def _fill(ws, global_w, specs, weights, biases):
    n, _, kw = ws.shape
    a = _lib.AffineBank()
    a.layers, a.n, a.kw, a.kg = len(specs), n, kw, 0 if global_w is None else int(global_w.shape[1])
    a.w_stride_n, a.w_stride_l = ws.stride(0), ws.stride(1)
    a.w, a.g = ws.data_ptr(), _lib.ptr(global_w)
    for i, (sp, w, b) in enumerate(zip(specs, weights, biases)):
        a.weight[i] = w.data_ptr()
        a.bias[i] = _lib.ptr(b)
        a.cout[i] = int(w.shape[0])
        a.w_index[i] = sp.w_index
        a.alpha[i] = float(sp.fc.weight_gain) * sp.scale
        a.beta[i] = float(sp.fc.bias_gain) * sp.scale
    return a


def _table(tensors):
    t = (C.c_void_p * len(tensors))()
    for i, x in enumerate(tensors):
        t[i] = None if x is None else x.data_ptr()
    return t


def supported(ws, global_w, specs):
    if not (ws.is_cuda and ws.dtype == torch.float32 and ws.ndim == 3 and ws.stride(2) == 1) or len(specs) == 0 or len(specs) > _lib.AFFINE_MAX:
        return False
    if global_w is not None and not (global_w.dtype == torch.float32 and global_w.is_contiguous() and global_w.shape[0] == ws.shape[0]):
        return False
    kg = 0 if global_w is None else global_w.shape[1]
    k = ws.shape[2] + kg
    for sp in specs:
        w, b = sp.fc.weight, sp.fc.bias
        if sp.fc.activation != 'linear' or w.dtype != torch.float32 or tuple(w.shape[1:]) != (k,) or not w.is_contiguous() or not w.is_cuda:
            return False
        if b is None or b.dtype != torch.float32 or not b.is_contiguous():
            return False
        if not 0 <= sp.w_index < ws.shape[1]:
            return False
    return (k <= 1536 and ws.shape[2] % 4 == 0 and kg % 4 == 0 and ws.stride(0) % 4 == 0 and ws.stride(1) % 4 == 0
            and ws.data_ptr() % 16 == 0 and (global_w is None or global_w.data_ptr() % 16 == 0))


class _AffineBank(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ws, global_w, specs, *params):
        nl = len(specs)
        weights, biases = params[:nl], params[nl:]
        lib = _lib.load()
        a = _fill(ws, global_w, specs, weights, biases)
        ys = [torch.empty([ws.shape[0], int(w.shape[0])], dtype=torch.float32, device=ws.device) for w in weights]
        rc = _lib.check(lib.afcm_affine_bank_fwd(a, _table(ys), _lib.stream_ptr(ws)), 'affine_bank')
        if rc == _lib.E_NOKERNEL:
            raise RuntimeError('affine_bank: shapes outside the kernels (check supported() first)')
        ctx.save_for_backward(ws, global_w, *params)
        ctx.specs = specs
        return tuple(ys)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        ws, global_w = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        specs = ctx.specs
        nl = len(specs)
        weights, biases = params[:nl], params[nl:]
        lib = _lib.load()
        a = _fill(ws, global_w, specs, weights, biases)
        gs = [None if g is None else g.contiguous() for g in gs]
        need_w = [ctx.needs_input_grad[3 + i] for i in range(nl)]
        need_b = [ctx.needs_input_grad[3 + nl + i] for i in range(nl)]
        dws = [torch.empty_like(w) if nw else None for w, nw in zip(weights, need_w)]
        dbs = [torch.empty_like(b) if nb else None for b, nb in zip(biases, need_b)]
        need_x = ctx.needs_input_grad[0] or (global_w is not None and ctx.needs_input_grad[1])
        d_ws_l = d_g = wsp = None
        if need_x:
            d_ws_l = torch.empty([ws.shape[0], nl, ws.shape[2]], dtype=torch.float32, device=ws.device)
            d_g = torch.empty_like(global_w) if global_w is not None else None
            wsp = torch.empty([int(lib.afcm_affine_bank_workspace_bytes(a))], dtype=torch.uint8, device=ws.device)
        any_w = any(need_w) or any(need_b)
        _lib.check(lib.afcm_affine_bank_bwd(a, _table(gs), _table(dws) if any_w else None, _table(dbs) if any_w else None, _lib.ptr(d_ws_l), _lib.ptr(d_g),
                                            _lib.ptr(wsp), _lib.stream_ptr(ws)), 'affine_bank')
        d_ws = None
        if ctx.needs_input_grad[0]:
            # the per-layer latent gradients back onto ws's latent axis (several layers may share a latent)
            idx = [sp.w_index for sp in specs]
            if idx == list(range(idx[0], idx[0] + nl)):
                if idx[0] == 0 and nl == ws.shape[1]:
                    d_ws = d_ws_l
                else:
                    d_ws = torch.zeros(ws.shape, dtype=ws.dtype, device=ws.device)
                    d_ws[:, idx[0]:idx[0] + nl] = d_ws_l
            else:
                d_ws = torch.zeros(ws.shape, dtype=ws.dtype, device=ws.device)
                d_ws.index_add_(1, torch.tensor(idx, device=ws.device), d_ws_l)
        return (d_ws, d_g if (global_w is not None and ctx.needs_input_grad[1]) else None, None) + tuple(dws) + tuple(dbs)


def affine_bank(ws, global_w, specs):
    """styles of every layer in `specs` ([N, cout_l] fp32 each).  ws: [N, L, kw] fp32 (any batch / latent strides that are multiples of
    4 elements), global_w: [N, kg] fp32 contiguous or None."""
    weights = [sp.fc.weight for sp in specs]
    biases = [sp.fc.bias for sp in specs]
    _lib.require_gpu(ws, global_w, *weights)
    return _AffineBank.apply(ws, global_w, tuple(specs), *(tuple(weights) + tuple(biases)))
