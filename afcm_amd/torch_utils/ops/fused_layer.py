"""One generator layer as ONE autograd node: conv (+ bias in its epilogue) -> filtered_lrelu (+ encoder skip, x next
layer's style factor in ITS epilogue), for 16-bit activations on the matrix-core kernels.

The reference composes a layer from separate ops (NET:366-377 / NET:503-511): modulated_conv2d (which multiplies the
activations by the styles and the result by the demodulation coefficients), filtered_lrelu (which adds the bias first),
``x + x_skip``.  Run op by op, each per-plane factor and the skip add is a full pass over the activations in HBM.  All of
them are per-plane scalars or elementwise adds around linear operators, so they fold into the epilogues of the two
kernels that touch the data anyway:

    forward   y = d[n,o] * conv(w_hat, xs) + b[o]                  (conv epilogue: demodulation + bias)
              z = (filtered_lrelu(y) + skip) * s_next[n,o]         (filtered_lrelu epilogue: skip + next layer's styles)
    backward  dys = (d * s_next)[n,o] * filtered_lrelu^T(g)        (the transposed op is linear in g given the sign codes)
              dxs = conv^T(w_hat, dys) (* s[n,i] when this node applied the styles itself),  dw_hat = wgrad(dys, xs)
              db[o] = sum_n planesum(dys)[n,o] / d[n,o]            (per-tile sums emitted by the backward kernel)
              dd    = (<dys, y> - b * planesum(dys)) / d^2,   ds_next = <g, z> / s_next,   dskip = g * s_next

``z`` comes out already multiplied by the next layer's styles, so the consumer is called with ``prescaled=True`` and this
node owns the gradient of those styles.  Same arithmetic as the op-by-op path up to 16-bit rounding (one rounding
fewer per fused factor).
"""
import torch

from ... import _lib
from . import _rows
from . import conv2d as _conv
from . import filtered_lrelu as _flr


class _ConvFilteredLRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, in_scale, out_scale, bias, fu, fd, skip, next_scale, conv_pad, cfg, prescaled, packed=None, fork=None, link_in=None, link_out=None):
        _lib.require_gpu(x, w, in_scale, out_scale, bias, fu, fd, skip, next_scale)
        cout, cin, ks, _ = w.shape
        xs = _conv.scale_planes(x, in_scale) if (in_scale is not None and not prescaled) else x
        ctx.wpt = None
        need_wpt = ctx.needs_input_grad[0] or (ctx.needs_input_grad[2] and not prescaled)
        if packed is not None and packed[0][0].dtype == x.dtype and (packed[1] is not None or not need_wpt):
            # both weight images come from the caller's multi-layer pack (conv2d.pack_weights_bank); the image must be THIS weight's
            (wp, rows_pad), ctx.wpt = packed
            bk = wp.shape[3]
            assert tuple(wp.shape) == ((cin + bk - 1) // bk, ks * ks, rows_pad, bk) and rows_pad == (cout + 63) // 64 * 64, \
                f'packed weight image {tuple(wp.shape)} does not belong to a {tuple(w.shape)} weight'
        elif need_wpt:
            (wp, rows_pad), ctx.wpt = _conv.pack_weights_both(w, x.dtype)      # the backward's weight image from the same launch
        else:
            wp, rows_pad = _conv.pack_weights(w, x.dtype, 0)
        # y, z (and dys, dx in backward) are row-pitched: rows on 64-byte boundaries (_rows.pitch_for), read and written in place by the kernels of this
        # node and of its neighbours (_rows.py)
        y = _conv._conv_raw(xs, wp, rows_pad, out_scale, cout, ks, conv_pad, obias=bias, pitched_out=True)
        need_grad = any(ctx.needs_input_grad[:5]) or ctx.needs_input_grad[7] or ctx.needs_input_grad[8]
        keep_y = out_scale is not None and ctx.needs_input_grad[3]
        flags = [] if (keep_y and need_grad and HOMOGENEOUS_DOT) else None
        z, signs, layout, _ = _flr._run(y, fu, fd, None, None, cfg, need_grad, oscale=next_scale, skip=skip, pitched_out=True, clamp_flags_out=flags)
        flags = flags[0] if flags else None
        # the demodulation gradient needs <dys, y>: with the flags it comes from <g, z> (and <g, skip>) by homogeneity, y is read
        # for flagged planes only (see backward); z is this node's output, keeping it costs nothing
        keep_z = (next_scale is not None and ctx.needs_input_grad[8]) or flags is not None
        ctx.save_for_backward(xs, w, in_scale, out_scale, bias, fu, fd, signs, next_scale, y if keep_y else None, z if keep_z else None,
                              flags, skip if (flags is not None and skip is not None) else None)
        # the skip tensor is one arm of a SkipFork: its gradient leaves this node UNSCALED (g itself, no pass over it) and the fork's backward
        # applies s_next while it adds the two arms (C ABI afcm_axpy_planes)
        ctx.skip_by_fork = False
        if fork is not None and skip is not None and next_scale is not None and ctx.needs_input_grad[7]:
            fork.scale = next_scale.detach()
            ctx.skip_by_fork = True
        # <g, z> of THIS node's output (its styles' gradient, the demodulation gradient by homogeneity) can come from the consumer's weight
        # gradient (LayerLink below): say so to the consumer; and remember the producer's request for this node's input
        ctx.link_out = link_out if (link_out is not None and next_scale is not None) else None
        if ctx.link_out is not None:
            ctx.link_out.want, ctx.link_out.gz = True, None
        ctx.link_in = link_in if (link_in is not None and prescaled) else None
        ctx.meta = (conv_pad, cfg, bool(prescaled), layout, tuple(y.shape), tuple(z.shape), skip is not None)
        return z

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        xs, w, in_scale, out_scale, bias, fu, fd, signs, next_scale, y, z, flags, skip = ctx.saved_tensors
        conv_pad, cfg, prescaled, layout, y_shape, z_shape, has_skip = ctx.meta
        cout, cin, ks, _ = w.shape
        f32 = torch.float32
        assert g.dtype in (torch.bfloat16, torch.float16)
        # dys = d * dL/dy: the transposed filtered_lrelu with both per-plane factors in its epilogue
        bcfg = _flr._backward_cfg(cfg, fu, fd, y_shape, z_shape, layout)
        dys, _, _, psum = _flr._run(g, fd, fu, None, signs, bcfg, False, want_plane_sum=True, oscale=out_scale, oscale2=next_scale, pitched_out=True)
        dx = dw = d_in = d_out = db = d_skip = d_next = None
        # bias / next-styles / demodulation gradients: two plane dot products + ONE small kernel (C ABI afcm_layer_bwd_coefs)
        n, o = int(g.shape[0]), int(g.shape[1])
        want_db, want_next, want_out = ctx.needs_input_grad[4], ctx.needs_input_grad[8], ctx.needs_input_grad[3]
        if want_db or want_next or want_out:
            lib = _lib.load()
            gz = None
            if want_next or (want_out and flags is not None):
                # from the consumer's backward (it ran before this one) where its weight gradient could supply it, else a pass over g and z
                if ctx.link_out is not None and ctx.link_out.gz is not None and tuple(ctx.link_out.gz.shape) == (n, o):
                    gz, ctx.link_out.gz = ctx.link_out.gz, None
                else:
                    gz = _conv.plane_dot(g, z)
            osc = None if out_scale is None else out_scale.to(f32).contiguous()
            nsc = None if next_scale is None else next_scale.to(f32).contiguous()
            if want_out and flags is not None:
                # <dys, y> = d <dL/dy, y> = d (<g, z> - s_next <g, skip>): without an active clamp the fused filtered_lrelu is positively
                # homogeneous of degree 1 (linear filters around a leaky ReLU), so J(y) y = F(y) = z / s_next - skip.  The forward
                # kernels flagged every strip whose activations could reach the clamp; a flagged plane gets the real dot product,
                # on the device (C ABI afcm_plane_dot_gated_ld) -- 14 full-plane dot products per step become 14 launches that read a
                # few flags per plane
                gsk = _conv.plane_dot(g, skip) if skip is not None else None
                dysy = _conv.plane_dot_gated(dys, y, flags, osc, gz, nsc, gsk)
            else:
                dysy = _conv.plane_dot(dys, y) if want_out else None
            db32 = torch.empty([o], dtype=f32, device=g.device) if want_db else None
            dn32 = torch.empty([n, o], dtype=f32, device=g.device) if want_next else None
            do32 = torch.empty([n, o], dtype=f32, device=g.device) if want_out else None
            b32 = None if bias is None else bias.to(f32).contiguous()
            _lib.check(lib.afcm_layer_bwd_coefs(_lib.ptr(db32), _lib.ptr(dn32), _lib.ptr(do32), psum.data_ptr(), int(psum.shape[2]), _lib.ptr(osc),
                                                _lib.ptr(nsc), _lib.ptr(b32), _lib.ptr(gz if want_next else None), _lib.ptr(dysy), n, o, _lib.stream_ptr(g)),
                       'layer_bwd_coefs')
            db = None if db32 is None else db32.to(bias.dtype)
            d_next = None if dn32 is None else dn32.to(next_scale.dtype)
            d_out = None if do32 is None else do32.to(out_scale.dtype)
        if has_skip and ctx.needs_input_grad[7]:
            d_skip = _conv.scale_planes(g, next_scale) if (next_scale is not None and not ctx.skip_by_fork) else g
        if ctx.needs_input_grad[0] or (ctx.needs_input_grad[2] and not prescaled):
            wpt, rows_pad = ctx.wpt if (ctx.wpt is not None and ctx.wpt[0].dtype == g.dtype) else _conv.pack_weights(w, g.dtype, 1)
            eff_in = None if prescaled else in_scale
            dx = _conv._conv_raw(dys, wpt, rows_pad, eff_in, cin, ks, ks - 1 - conv_pad, pitched_out=True)
            if ctx.needs_input_grad[2] and eff_in is not None:
                s2 = in_scale.to(f32).square()
                d_in = torch.where(s2 > 0, _conv.plane_dot(xs, dx) / s2.clamp_min(1e-30), torch.zeros_like(s2)).to(in_scale.dtype)
        if ctx.needs_input_grad[1]:
            if ctx.link_in is not None and ctx.link_in.want:
                # ... and with it <xs, dx> per input plane = the producer's <g, z> (see LayerLink)
                dw, ctx.link_in.gz = _conv._wgrad_raw(dys, xs, cout, cin, ks, conv_pad, dots_with=w)
                dw = dw.to(w.dtype)
            else:
                dw = _conv._wgrad_raw(dys, xs, cout, cin, ks, conv_pad).to(w.dtype)
        return dx, dw, d_in, d_out, db, None, None, d_skip, d_next, None, None, None, None, None, None, None


# the demodulation gradient's <dys, y> from <g, z> where no strip of a plane could reach the clamp (tests switch it off to compare)
HOMOGENEOUS_DOT = True



def _cfg(up, down, padding, gain, slope, clamp):
    px0, px1, py0, py1 = _flr._parse_padding(padding)
    return (int(up), int(down), px0, px1, py0, py1, float(gain), float(slope), float(clamp if clamp is not None else 'inf'), False, 0, 0, 0)


def available(x, w, fu, fd, up, down, padding, gain, slope, clamp, conv_pad):
    """The fused node needs the matrix-core filtered_lrelu (16-bit activations, 12/24-tap separable filters, even widths)
    behind a 3x3 conv."""
    if w.shape[2] != 3 or x.dtype not in (torch.bfloat16, torch.float16) or x.device.type != 'cuda':
        return False
    yshape = [x.shape[0], w.shape[0], x.shape[2] + 2 * conv_pad - 2, x.shape[3] + 2 * conv_pad - 2]
    return _flr.matrix_core_available(yshape, x.dtype, x.device, fu, fd, _cfg(up, down, padding, gain, slope, clamp))


class LayerLink:
    """Between a producer layer that multiplied its output z by the consumer's styles (`next_scale`, consumer called with `prescaled`) and
    that consumer: the producer needs <g, z> per plane in its backward (g = dL/dz); the consumer's backward -- which runs first -- has the
    same numbers in its weight gradient's per-image slabs (<xs, dx> with xs = z, dx = g: conv2d._wgrad_raw(dots_with=...)), PROVIDED z
    has no other consumer.  The producer sets `want` in its forward, the consumer leaves `gz` ([N, C] fp32, or None where its split plan
    does not allow it) in its backward, the producer takes it or falls back to a pass over g and z."""
    __slots__ = ('want', 'gz')

    def __init__(self):
        self.want, self.gz = False, None


def conv_filtered_lrelu(x, w, in_scale, out_scale, bias, fu, fd, up, down, padding, gain, slope, clamp, conv_pad, skip=None,
                        next_scale=None, prescaled=False, packed=None, link_in=None, link_out=None):
    """z = (filtered_lrelu(out_scale * conv(w, in_scale * x) + bias; fu, fd, up, down, padding, gain, slope, clamp) + skip)
    * next_scale.  ``prescaled``: x already carries in_scale (the producer's epilogue applied it).  ``packed``: the two weight
    images of `w` from ``conv2d.pack_weights_bank`` (the caller packed several layers in one launch), else they are made here.
    ``link_in`` / ``link_out``: the LayerLink objects shared with the producer of x / the consumer of the result (x / the result must
    have NO other consumer)."""
    cfg = _cfg(up, down, padding, gain, slope, clamp)
    fork = getattr(skip, '_afcm_fork', None) if skip is not None else None
    return _ConvFilteredLRelu.apply(x, w, in_scale, out_scale, bias, fu, fd, skip, next_scale, int(conv_pad), cfg, bool(prescaled), packed, fork,
                                    link_in, link_out)


# ---- an encoder feature map that feeds the next encoder layer AND a decoder layer's skip input (NET:678-681, 371-377) ----------------------
class _ForkState:
    """What the decoder side tells the fork between forward and backward: the per-plane factor its epilogue applied to (F(y) + skip)."""
    __slots__ = ('scale',)

    def __init__(self):
        self.scale = None


class _SkipFork(torch.autograd.Function):
    """x -> (x, x).  Its backward is the accumulation autograd would do for a tensor with two consumers -- ga + gb -- as ONE pass that also
    applies the skip arm's pending style factor: ga + scale[n, c] * gb (C ABI afcm_axpy_planes).  Op by op that was scale_planes (a new
    156 MB tensor at 276^2) followed by autograd's add: five passes over the planes instead of three, two roundings instead of one."""

    @staticmethod
    def forward(ctx, x, state):
        ctx.state = state
        return x.view_as(x), x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ga, gb):
        scale = ctx.state.scale
        ctx.state.scale = None
        if gb is None:
            return ga, None
        if ga is None:
            return (gb if scale is None else _conv.scale_planes(gb, scale)), None
        out = None
        if ga.is_cuda and ga.dtype in (torch.bfloat16, torch.float16) and ga.dtype == gb.dtype and ga.shape == gb.shape:
            fa, fb = (_rows.whole_buffer(ga), _rows.whole_buffer(gb)) if not ga.is_contiguous() else (ga, gb if gb.is_contiguous() else None)
            if fa is not None and fb is not None and fa.shape == fb.shape:
                n, c, h, ld = fa.shape
                y = torch.empty_like(fa)
                sc = None if scale is None else scale.to(torch.float32).contiguous()
                rc = _lib.check(_lib.load().afcm_axpy_planes(y.data_ptr(), fa.data_ptr(), fb.data_ptr(), _lib.ptr(sc), _lib.dtype_code(fa), n * c, h * ld,
                                                            _lib.stream_ptr(fa)), 'axpy_planes')
                if rc == 0:
                    out = y[..., :ga.shape[3]]
        if out is None:
            out = ga + (gb if scale is None else _conv.scale_planes(gb, scale))
        return out, None


def skip_fork(x):
    """(x for the next encoder layer, x for E_features): two views of `x` whose gradients meet in one fused pass (see _SkipFork).  The
    second carries the fork's state as a Python attribute; `conv_filtered_lrelu(skip=...)` finds it there."""
    if not (x.requires_grad and torch.is_grad_enabled()):
        return x, x
    state = _ForkState()
    a, b = _SkipFork.apply(x, state)
    b._afcm_fork = state
    return a, b
