"""filtered_lrelu: bias -> upsample FIR -> leaky ReLU (gain, clamp) -> downsample FIR, fused on the GPU.

Drop-in for the reference's ``filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0,
gain=sqrt(2), slope=0.2, clamp=None, flip_filter=False, impl='cuda')`` (SG3OPS/filtered_lrelu.py:56-116).

``impl='cuda'`` on a ROCm tensor runs the HIP kernel ``afcm_filtered_lrelu``.  As in the reference
(SG3OPS/filtered_lrelu.py:159-272) the forward pass keeps only a 2-bit-per-element sign/clamp tensor,
and the backward pass is the same op with up/down and the filters swapped, reading those codes; so
``dx`` is again differentiable.  Parameter combinations without a fused kernel (non-separable filters,
other tap counts) take the generic path upfirdn2d -> in-place activation kernel -> upfirdn2d, also on
the GPU (SG3OPS/filtered_lrelu.py:223-229) -- never a CPU path.
"""
import collections
import warnings

import numpy as np
import torch

from ... import _lib
from ... import profiling
from . import _rows
from . import upfirdn2d as _ufd


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and 1 <= f.ndim <= 2
    return int(f.shape[-1]), int(f.shape[0])  # width, height


def _parse_padding(padding):
    if isinstance(padding, (int, np.integer)):
        padding = [int(padding)] * 2
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(v, (int, np.integer)) for v in padding)
    padding = [int(v) for v in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _filter_arg(f, device):
    """(pointer tensor or None, fw, fh-or-0) for the C ABI: fh == 0 marks a separable filter."""
    if f is None:
        return None, 1, 1
    if f.dtype != torch.float32:
        raise RuntimeError('fu and fd must be float32')
    if f.device != device:
        raise RuntimeError('all input tensors must reside on the same device')
    if f.numel() == 0:
        raise RuntimeError('fu and fd must not be empty')
    f = f.contiguous()
    return f, int(f.shape[-1]), (int(f.shape[0]) if f.ndim == 2 else 0)


def _act_inplace(y, si, sx, sy, gain, slope, clamp, write_signs):
    """In-place gain / leaky ReLU / clamp with sign handling (replaces `_plugin.filtered_lrelu_act_`,
    SG3OPS/filtered_lrelu.cpp:213-290).  Returns the sign tensor it wrote, or None."""
    lib = _lib.load()
    n, c, h, w = y.shape
    so = None
    mode = _lib.SIGNS_NONE
    s = None
    if write_signs:
        so = s = torch.empty([n, c, h, ((w + 15) & ~15) >> 2], dtype=torch.uint8, device=y.device)
        mode = _lib.SIGNS_WRITE
    elif si is not None:
        s = si
        mode = _lib.SIGNS_READ
    rc = lib.afcm_filtered_lrelu_act(_lib.ptr(y), _lib.ptr(s), _lib.dtype_code(y), n, c, h, w,
                                     0 if s is None else s.shape[2], 0 if s is None else s.shape[3], sx, sy,
                                     gain, slope, clamp, mode, _lib.stream_ptr(y))
    _lib.check(rc, 'filtered_lrelu_act_')
    return so


# layer configuration -> prepared constant-fragment buffer (or None: no matrix-core kernel).  Bounded, least recently used out
# first: an entry pins its 18 KB workspace and its two filter tensors, and a program that keeps building filters (a sweep, a
# long-lived server) must not grow it without limit.  A network holds ~30 distinct configurations (forward + transposed).
_workspaces = collections.OrderedDict()
_WORKSPACE_CACHE_ENTRIES = 512


def _mfma_workspace(a, fu_t, fd_t, x):
    """Constant Toeplitz fragments of the matrix-core kernels, built once per layer configuration
    (C ABI afcm_filtered_lrelu_prepare).  Returns a device tensor or None."""
    if x.dtype not in (torch.bfloat16, torch.float16) or fu_t is None or fd_t is None:
        return None
    # a READ call on codes of the wave kernels (layout 2) gets fragments whose rows are moved so that its strips fall on the sign
    # tensor's row blocks (csrc/filtered_lrelu_mfma.hip wave_read_origin): the shift depends on sy mod 16
    aligned_read = (a.sy % 16) if (a.sign_mode == _lib.SIGNS_READ and a.sign_layout == 2) else None
    key = (x.device, x.dtype, fu_t.data_ptr(), fu_t._version, fd_t.data_ptr(), fd_t._version, a.fuw, a.fuh, a.fdw, a.fdh,
           a.up, a.down, a.px0, a.py0, a.gain, a.slope, a.flip_filter, aligned_read)
    if key not in _workspaces:
        lib = _lib.load()
        ws = torch.empty([lib.afcm_filtered_lrelu_workspace_bytes()], dtype=torch.uint8, device=x.device)
        a.workspace = ws.data_ptr()
        a.fu, a.fd = fu_t.data_ptr(), fd_t.data_ptr()
        rc = _lib.check(lib.afcm_filtered_lrelu_prepare(a, _lib.stream_ptr(x)), 'filtered_lrelu_prepare')
        # keep the filters alive with the workspace: the key holds their addresses
        # (a negative entry pins its filters too: a recycled address must not inherit "no matrix-core kernel")
        _workspaces[key] = (ws if rc == 0 else None, fu_t, fd_t)
        while len(_workspaces) > _WORKSPACE_CACHE_ENTRIES:
            _workspaces.popitem(last=False)       # the caching allocator keeps the block until queued kernels have run
    else:
        _workspaces.move_to_end(key)
    return _workspaces[key][0]


class NoFusedKernel(Exception):
    """Raised by _run(no_fallback=True) where the reference plugin returns return_code -1 (filtered_lrelu.cpp:52-56)."""


def _run(x, fu, fd, b, si, cfg, write_signs, want_plane_sum=False, oscale=None, skip=None, oscale2=None, allow_mfma=True, no_fallback=False,
         pitched_out=False, clamp_flags_out=None):
    """One launch of the op (C ABI afcm_filtered_lrelu, or the generic GPU path when there is no fused kernel).
    Returns (y, signs written or None, sign layout, per-plane output sums or None).  x / skip may be row-pitched views (_rows.py):
    kernels that take a pitch read them in place, the others get a contiguous copy; ``pitched_out``: y comes back row-pitched
    when the selected kernel can write one (callers that hand y to pitch-aware kernels only); ``clamp_flags_out``: a list that
    receives the per-strip "could reach the clamp" flags (int32 [N, C, slots]) of a sign-writing call of the wave kernels (C ABI
    afcm_filtered_lrelu_args.clamp_flags) -- left empty when another kernel family runs."""
    up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter, sx, sy, si_layout = cfg
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    _lib.require_gpu(x, fu, fd, b, si)
    lib = _lib.load()
    if x.numel() == 0:
        raise RuntimeError('x is empty')
    x, xld = _rows.rows(x)
    if b is not None:
        if b.dtype != x.dtype:
            raise RuntimeError('x and b must have the same dtype')
        if b.ndim != 1 or b.shape[0] != x.shape[1]:
            raise RuntimeError('b must be a vector with the same number of channels as x')
        b = b.contiguous()
    fu_t, fuw, fuh = _filter_arg(fu, x.device)
    fd_t, fdw, fdh = _filter_arg(fd, x.device)

    a = _lib.FilteredLReluArgs()
    a.dtype = _lib.dtype_code(x)
    a.n, a.c, a.xh, a.xw = x.shape
    a.fuw, a.fuh, a.fdw, a.fdh = fuw, fuh, fdw, fdh
    a.up, a.down = up, down
    a.px0, a.px1, a.py0, a.py1 = px0, px1, py0, py1
    a.sx, a.sy = sx, sy
    a.gain, a.slope, a.clamp = gain, slope, clamp
    a.flip_filter = int(flip_filter)
    a.sign_mode = _lib.SIGNS_WRITE if write_signs else (_lib.SIGNS_READ if si is not None else _lib.SIGNS_NONE)
    # 16-bit activations: matrix-core kernels (signs in the row-quad layout); a given sign tensor fixes the family
    ws = None
    a.sign_layout = si_layout if si is not None else 0      # (prepare: a READ call's fragments depend on the layout it reads)
    if allow_mfma and (si is None or si_layout in (1, 2)):
        ws = _mfma_workspace(a, fu_t, fd_t, x)
        if si is not None and ws is None:
            raise RuntimeError('filtered_lrelu: sign tensor was written by the matrix-core kernels but this call has none')
    a.workspace = _lib.ptr(ws)
    a.sign_layout = si_layout if si is not None else 0
    a.b = _lib.ptr(b)                     # the kernel family (and with it the sign layout a WRITE call produces) depends on it
    a.x_pitch = xld if xld != x.shape[3] else 0         # (the kernel family may depend on the plane size in memory)
    _lib.check(lib.afcm_filtered_lrelu_shapes(a), 'filtered_lrelu')
    kld = 0
    if skip is not None:
        skip, kld = _rows.rows(skip)
    if not a.row_pitch_ok:                               # this call's kernel takes dense tensors
        x, a.x_pitch = _rows.dense(x), 0
        if skip is not None:
            skip, kld = _rows.dense(skip), 0
    y = _rows.empty([a.n, a.c, a.yh, a.yw], x.dtype, x.device, pitched=bool(pitched_out and a.row_pitch_ok))
    a.y_pitch = 0 if y.is_contiguous() else y.stride(2)
    so = None
    if write_signs:
        so = torch.empty([a.n, a.c, a.sh, a.swb], dtype=torch.uint8, device=x.device)
        a.signs = so.data_ptr()
    elif si is not None:
        if si.dtype != torch.uint8 or si.ndim != 4 or not si.is_contiguous() or si.shape[:2] != x.shape[:2]:
            raise RuntimeError('signs must be a contiguous uint8 tensor with the same batch & channels as x')
        a.sh, a.swb = si.shape[2], si.shape[3]
        a.signs = si.data_ptr()
    a.x, a.y, a.b = x.data_ptr(), y.data_ptr(), _lib.ptr(b)
    if oscale is None and oscale2 is not None:
        oscale, oscale2 = oscale2, None
    if oscale is not None or skip is not None:
        # epilogue factors of the fused layer op (matrix-core kernels only; the C side rejects anything else)
        if ws is None:
            raise RuntimeError('filtered_lrelu: oscale / skip need the matrix-core kernels')
        if oscale is not None:
            oscale = oscale.to(torch.float32).contiguous()
            assert oscale.numel() == a.n * a.c
        if skip is not None:
            assert skip.dtype == x.dtype and tuple(skip.shape) == (a.n, a.c, a.yh, a.yw)
            a.skip_pitch = kld if kld != a.yw else 0
        if oscale2 is not None:
            oscale2 = oscale2.to(torch.float32).contiguous()
            assert oscale2.numel() == a.n * a.c
        a.oscale, a.skip, a.oscale2 = _lib.ptr(oscale), _lib.ptr(skip), _lib.ptr(oscale2)
    psum = None
    if ws is not None and want_plane_sum and a.plane_sum_slots > 0:
        psum = torch.empty([a.n, a.c, a.plane_sum_slots], dtype=torch.float32, device=x.device)   # every slot is written
        a.plane_sum = psum.data_ptr()
    flags = None
    if clamp_flags_out is not None and ws is not None and write_signs and b is None and a.plane_sum_slots > 0:
        flags = torch.empty([a.n, a.c, a.plane_sum_slots], dtype=torch.int32, device=x.device)       # every slot is written (layout 2 only: checked below)
        a.clamp_flags = flags.data_ptr()
    a.fu, a.fd = _lib.ptr(fu_t), _lib.ptr(fd_t)
    span = profiling.span('filtered_lrelu', (x.numel() + y.numel()) * x.element_size()
                          + (so.numel() if so is not None else (si.numel() if si is not None else 0)))
    rc = _lib.check(lib.afcm_filtered_lrelu(a, _lib.stream_ptr(x)), 'filtered_lrelu')
    if span is not None:
        span.end()
    layout = a.sign_layout
    if flags is not None and layout == 2 and rc == 0:
        clamp_flags_out.append(flags)

    if rc == _lib.E_NOKERNEL and no_fallback:
        raise NoFusedKernel()
    if rc == _lib.E_NOKERNEL:
        # Generic path, still on the GPU and still keeping only the packed signs for backward.
        warnings.warn('filtered_lrelu called with parameters that have no fused HIP kernel, using generic fallback', RuntimeWarning)
        y = x if b is None else x + b.reshape(1, -1, 1, 1)
        y = _ufd._forward_raw(y, fu, (up, up), (1, 1), (px0, px1, py0, py1), flip_filter, float(up ** 2))
        if y is x:
            y = y.clone()
        so = _act_inplace(y, si, sx, sy, gain, slope, clamp, write_signs)
        y = _ufd._forward_raw(y, fd, (1, 1), (down, down), (0, 0, 0, 0), flip_filter, 1.0)
        layout, psum = 0, None
    return y, so, layout, psum


def _backward_cfg(cfg, fu, fd, x_shape, y_shape, sign_layout):
    """Configuration of the transposed op (SG3OPS/filtered_lrelu.py:252-263): swap the resampling roles, flip the filters,
    drop the clamp (the codes already carry it) and shift the sign window."""
    up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter, sx, sy, _ = cfg
    _, _, xh, xw = x_shape
    _, _, yh, yw = y_shape
    fuw, fuh = _get_filter_size(fu)
    fdw, fdh = _get_filter_size(fd)
    if fu is not None and fu.ndim == 1:
        fuh = fuw
    if fd is not None and fd.ndim == 1:
        fdh = fdw
    pp = ((fuw - 1) + (fdw - 1) - px0, xw * up - yw * down + px0 - (up - 1),
          (fuh - 1) + (fdh - 1) - py0, xh * up - yh * down + py0 - (up - 1))
    gg = gain * (up ** 2) / (down ** 2)
    return (down, up) + pp + (gg, slope, float('inf'), not flip_filter, sx - (fuw - 1) + px0, sy - (fuh - 1) + py0, sign_layout)


class _FilteredLRelu(torch.autograd.Function):
    """x, fu, fd, b, si are tensors (or None); cfg carries the scalars of one call."""

    @staticmethod
    def forward(ctx, x, fu, fd, b, si, cfg):
        if si is not None and si.numel() == 0:
            si = None
        write_signs = si is None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[3])
        y, so, layout, _ = _run(x, fu, fd, b, si, cfg, write_signs)
        ctx.save_for_backward(fu, fd, si if si is not None else so)
        ctx.cfg = cfg
        ctx.sign_layout = layout
        ctx.x_shape = x.shape
        ctx.y_shape = y.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        fu, fd, si = ctx.saved_tensors
        assert not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2] or ctx.needs_input_grad[4])
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
            cfg = _backward_cfg(ctx.cfg, fu, fd, ctx.x_shape, ctx.y_shape, ctx.sign_layout)
            if torch.is_grad_enabled():
                # a higher-order graph is being recorded: stay differentiable, reduce the bias gradient with torch
                dx = _FilteredLRelu.apply(dy, fd, fu, None, si, cfg)
                psum = None
            else:
                dx, _, _, psum = _run(dy, fd, fu, None, si, cfg, False, want_plane_sum=bool(ctx.needs_input_grad[3]))
            if ctx.needs_input_grad[3]:
                # db = dx.sum([0, 2, 3]) (SG3OPS/filtered_lrelu.py:266); the matrix-core kernels already summed each plane
                if psum is not None:
                    db = psum.sum([0, 2]).to(dx.dtype)
                elif dx.requires_grad:
                    db = dx.sum([0, 2, 3])
                else:
                    # one workgroup per plane (C ABI afcm_plane_dot), then N x C -> C: 120 us against 170 for the framework's strided
                    # reduction on the generator's fp32 planes
                    from . import conv2d as _conv
                    db = _conv.plane_dot(dx).sum(0).to(dx.dtype)
        return dx, None, None, db, None, None


def _filtered_lrelu_generic(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter):
    """impl='ref' on a GPU tensor: the op as its definition composes it -- bias, upsampling FIR with gain up^2, leaky ReLU with
    gain / clamp, downsampling FIR (SG3OPS/filtered_lrelu.py:121-153) -- on the HIP upfirdn2d and bias_act kernels.  Unfused and
    memory-hungry like the reference's 'ref' twin (it keeps the up^2-times larger intermediate), arbitrarily differentiable;
    nothing here touches the CPU or oracle/."""
    from . import bias_act as _ba
    px0, px1, py0, py1 = _parse_padding(padding)
    y = _ba.bias_act(x=x, b=b)
    y = _ufd.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    y = _ba.bias_act(x=y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return _ufd.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    r"""Filtered leaky ReLU for a batch of 2-D images; see the module docstring.

    Args follow SG3OPS/filtered_lrelu.py:85-111: `x` `[N, C, H, W]` float32/float16 (bfloat16 is
    additionally supported here), `fu`/`fd` float32 `[taps]` (separable), `[fh, fw]` or `None`,
    `b` `[C]` of `x`'s dtype or `None`, integer `up`/`down`, `padding` int, `[x, y]` or
    `[x_before, x_after, y_before, y_after]` on the upsampled grid (negative = crop), `gain`,
    `slope`, `clamp` (None = off), `flip_filter` (False = convolution).  Returns `[N, C, H', W']`.
    """
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        _lib.require_gpu(x, fu, fd, b)
        return _filtered_lrelu_generic(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter)
    assert isinstance(up, (int, np.integer)) and up >= 1
    assert isinstance(down, (int, np.integer)) and down >= 1
    px0, px1, py0, py1 = _parse_padding(padding)
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    clamp = float(clamp if clamp is not None else 'inf')
    _lib.require_gpu(x, fu, fd, b)
    cfg = (int(up), int(down), px0, px1, py0, py1, float(gain), float(slope), clamp, bool(flip_filter), 0, 0, 0)
    return _FilteredLRelu.apply(x, fu, fd, b, None, cfg)


def matrix_core_available(shape, dtype, device, fu, fd, cfg):
    """True when a call on an input of this shape / dtype would run on the matrix-core kernels (16-bit dtype, separable
    12/24-tap case, even widths): the condition for the epilogue factors of the fused layer op."""
    if dtype not in (torch.bfloat16, torch.float16) or fu is None or fd is None or fu.ndim != 1 or fd.ndim != 1:
        return False
    up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter = cfg[:10]
    a = _lib.FilteredLReluArgs()
    a.dtype = _lib._DTYPES[dtype]
    a.n, a.c, a.xh, a.xw = [int(v) for v in shape]
    a.fuw, a.fuh, a.fdw, a.fdh = int(fu.shape[0]), 0, int(fd.shape[0]), 0
    a.up, a.down = up, down
    a.px0, a.px1, a.py0, a.py1 = px0, px1, py0, py1
    a.gain, a.slope, a.clamp = gain, slope, clamp
    a.flip_filter = int(flip_filter)
    a.sign_mode = _lib.SIGNS_WRITE
    return _mfma_workspace(a, fu.contiguous(), fd.contiguous(), torch.empty(0, dtype=dtype, device=device)) is not None
