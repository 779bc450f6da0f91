"""Equalised-learning-rate dense layers as ONE launch per layer and direction (C ABI afcm_fc_act_fwd / _bwd) and the mapping
network's input stage as one launch each way (afcm_mapping_input_fwd / _bwd).

The reference composes ``FullyConnectedLayer.forward`` (NET:97-104) from ``addmm`` / ``matmul`` and the ``bias_act`` plugin;
``MappingNetwork.forward`` (NET:143-157) adds two normalisations, the embedding FC and a ``cat`` in front of eight such layers.
Run op by op on the GPU that is ~100 launches of 3-5 us per training step (forward + backward) for ~0.1 GFLOP.  Here a layer
is one kernel forward (GEMM + bias + leaky ReLU) and one backward (activation gradient from the saved output, both GEMMs and
the bias column sum); arithmetic is exact fp32 (the f32 matrix instruction = an fmaf chain), so results differ from the GEMM
library's only by summation order.

First-order gradients only (``once_differentiable``): the generator's dense layers are never inside a double backward (the R1
penalty differentiates the DISCRIMINATOR twice, which has its own layer class).  ``ENABLED = False`` restores the op-by-op
composition (tests compare the two).
"""
import torch

from ... import _lib

ENABLED = True
_ACTS = {'linear': 0, 'lrelu': 1}
_MAPIN_WS = {}


def supported(x, weight, activation):
    return (ENABLED and activation in _ACTS and x.ndim == 2 and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and 0 < x.shape[0] <= 64 and weight.shape[1] % 16 == 0 and weight.shape[0] % 16 == 0)


class _FcAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, alpha, beta, act):
        _lib.require_gpu(x, w, b)
        lib = _lib.load()
        x, w = x.contiguous(), w.contiguous()
        b = None if b is None else b.contiguous()
        n, cin = x.shape
        cout = int(w.shape[0])
        y = torch.empty([n, cout], dtype=torch.float32, device=x.device)
        rc = _lib.check(lib.afcm_fc_act_fwd(y.data_ptr(), x.data_ptr(), w.data_ptr(), _lib.ptr(b), n, cin, cout, float(alpha), float(beta), int(act),
                                            _lib.stream_ptr(x)), 'fc_act_fwd')
        if rc != 0:
            raise RuntimeError('fc_act_fwd: no kernel for this shape (check fc_bank.supported first)')
        ctx.save_for_backward(x, w, y if act else None)
        ctx.cfg = (float(alpha), float(beta), int(act), b is not None)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        alpha, beta, act, has_b = ctx.cfg
        lib = _lib.load()
        gy = gy.contiguous()
        n, cin = x.shape
        cout = int(w.shape[0])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        db = torch.empty([cout], dtype=torch.float32, device=x.device) if (has_b and ctx.needs_input_grad[2]) else None
        rc = _lib.check(lib.afcm_fc_act_bwd(_lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), gy.data_ptr(), _lib.ptr(y), x.data_ptr(), w.data_ptr(), n, cin, cout,
                                            alpha, beta, act, _lib.stream_ptr(x)), 'fc_act_bwd')
        if rc != 0:
            raise RuntimeError('fc_act_bwd: no kernel for this shape')
        return dx, dw, db, None, None, None


def fc_act(x, weight, bias, weight_gain, bias_gain, activation):
    """act(weight_gain * x @ weight.T + bias_gain * bias) for a 2-D fp32 `x` (check ``supported`` first)."""
    return _FcAct.apply(x, weight, bias, float(weight_gain), float(bias_gain), _ACTS[activation])


class _MappingInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, c, ew, eb, alpha, beta):
        _lib.require_gpu(z, c, ew, eb)
        lib = _lib.load()
        z = z.contiguous()
        n, zdim = z.shape
        cdim = 0 if c is None else int(c.shape[1])
        wdim = 0 if ew is None else int(ew.shape[0])
        if cdim:
            c, ew = c.contiguous(), ew.contiguous()
            eb = None if eb is None else eb.contiguous()
        x0 = torch.empty([n, zdim + (wdim if cdim else 0)], dtype=torch.float32, device=z.device)
        _lib.check(lib.afcm_mapping_input_fwd(x0.data_ptr(), z.data_ptr(), _lib.ptr(c) if cdim else None, _lib.ptr(ew) if cdim else None,
                                              _lib.ptr(eb) if cdim else None, n, zdim, cdim, wdim, float(alpha), float(beta), _lib.stream_ptr(z)),
                   'mapping_input_fwd')
        ctx.save_for_backward(c, ew, eb)
        ctx.cfg = (n, zdim, cdim, wdim, float(alpha), float(beta))
        return x0

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx0):
        c, ew, eb = ctx.saved_tensors
        n, zdim, cdim, wdim, alpha, beta = ctx.cfg
        if cdim == 0 or not (ctx.needs_input_grad[2] or ctx.needs_input_grad[3]):
            return None, None, None, None, None, None
        lib = _lib.load()
        gx0 = gx0.contiguous()
        dew = torch.empty_like(ew)
        deb = torch.empty_like(eb) if (eb is not None and ctx.needs_input_grad[3]) else None
        key = (n, wdim, gx0.device)
        ws = _MAPIN_WS.get(key)
        if ws is None:
            # (zeroed once: the kernel leaves its ticket word at zero; one launch at a time per device and shape -- launches on one stream)
            ws = _MAPIN_WS[key] = torch.zeros([lib.afcm_mapping_input_bwd_workspace_bytes(n, wdim) // 4], dtype=torch.float32, device=gx0.device)
        _lib.check(lib.afcm_mapping_input_bwd(dew.data_ptr(), _lib.ptr(deb), gx0.data_ptr(), c.data_ptr(), ew.data_ptr(), _lib.ptr(eb), n, zdim, cdim, wdim,
                                              alpha, beta, ws.data_ptr(), _lib.stream_ptr(gx0)), 'mapping_input_bwd')
        return None, None, (dew if ctx.needs_input_grad[2] else None), deb, None, None


def mapping_input_supported(z, c, embed):
    return (ENABLED and z.is_cuda and z.dtype == torch.float32 and z.ndim == 2 and not z.requires_grad
            and (c is None or (c.dtype == torch.float32 and not c.requires_grad and embed is not None and embed.activation == 'linear')))


def mapping_input(z, c, embed):
    """cat(normalize(z), normalize(embed(c))) (NET:143-150) as one launch; `embed` is the FullyConnectedLayer c_dim -> w_dim or None."""
    if c is None or embed is None:
        return _MappingInput.apply(z, None, None, None, 1.0, 1.0)
    return _MappingInput.apply(z, c, embed.weight, embed.bias, float(embed.weight_gain), float(embed.bias_gain))
