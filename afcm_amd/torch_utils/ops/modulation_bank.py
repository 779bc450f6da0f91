"""The weight-side half of modulated_conv2d for a LIST of layers as ONE op (C ABI afcm_modulation_bank_*, csrc/modulation.hip).

``modulation_coefficients_fused`` (conv2d.py) turns one layer's (weight, raw styles, magnitude EMA) into (w_hat, in_scale, out_scale)
in 2 launches forward and 3 backward (NET:41-57 with input_gain = magnitude_ema.rsqrt(), NET:346).  The decoder has 15 such layers whose
styles all exist before the first activation is touched, so ``modulation_bank`` does them together: 2 + 3 launches for the whole list,
the same kernel bodies (bit-identical results), and one allocation each for the internal tensors.  First-order only
(``once_differentiable``), like the per-layer ops."""
import torch

from ... import _lib


class Item:
    """One layer: conv weight [O, I, k, k], raw styles [N, I], the magnitude EMA buffer (or None) and whether it demodulates."""

    def __init__(self, weight, styles, magnitude, demodulate):
        self.weight, self.styles, self.magnitude, self.demodulate = weight, styles, magnitude, bool(demodulate)


def supported(items):
    if not 0 < len(items) <= _lib.MODULATION_MAX:
        return False
    n = items[0].styles.shape[0]
    for it in items:
        w, t = it.weight, it.styles
        if not (w.is_cuda and t.is_cuda and w.dtype == torch.float32 and t.dtype == torch.float32 and w.ndim == 4 and t.ndim == 2):
            return False
        if t.shape[0] != n or t.shape[1] != w.shape[1] or max(w.shape[0], w.shape[1]) > 16384:
            return False
    return True


class _ModulationBank(torch.autograd.Function):
    """inputs: demods (tuple of bool), then the demodulating layers' weights, every layer's styles, every layer's magnitude (or None);
    outputs: w_hat of the demodulating layers, s_eff of every layer, d of the demodulating layers."""

    @staticmethod
    def forward(ctx, demods, *tensors):
        lib = _lib.load()
        nl, nd = len(demods), sum(demods)
        weights = [w.detach().contiguous() for w in tensors[:nd]]
        styles = [t.detach().contiguous() for t in tensors[nd:nd + nl]]
        mags = [None if m is None else m.detach().to(torch.float32).reshape(1) for m in tensors[nd + nl:]]
        dev = styles[0].device
        n = int(styles[0].shape[0])
        # internal tensors (wsq, scale per demodulating layer; r per layer) from one allocation
        sizes = [int(w.shape[0]) * int(w.shape[1]) + int(w.shape[0]) for w in weights]
        flat = torch.empty([sum(sizes) + nl], dtype=torch.float32, device=dev)
        table = (_lib.ModulationLayer * nl)()
        w_hats, s_effs, ds = [], [], []
        off, k = 0, 0
        for l in range(nl):
            L, t = table[l], styles[l]
            L.cin, L.demodulate = int(t.shape[1]), int(demods[l])
            L.t, L.magnitude = t.data_ptr(), _lib.ptr(mags[l])
            s_eff = torch.empty_like(t)
            s_effs.append(s_eff)
            L.s_eff = s_eff.data_ptr()
            L.r = flat.data_ptr() + 4 * (sum(sizes) + l)
            if demods[l]:
                w = weights[k]
                o, i, kh, kw = w.shape
                L.cout, L.kk = int(o), int(kh * kw)
                w_hat = torch.empty_like(w)
                d = torch.empty([n, o], dtype=torch.float32, device=dev)
                w_hats.append(w_hat)
                ds.append(d)
                L.w, L.w_hat, L.d = w.data_ptr(), w_hat.data_ptr(), d.data_ptr()
                L.wsq = flat.data_ptr() + 4 * off
                L.scale = flat.data_ptr() + 4 * (off + o * i)
                off += sizes[k]
                k += 1
        _lib.check(lib.afcm_modulation_bank_fwd(table, nl, n, _lib.stream_ptr(styles[0])), 'modulation_bank_fwd')
        ctx.save_for_backward(flat, *w_hats, *styles, *ds, *[m for m in mags if m is not None])
        ctx.demods, ctx.has_mag, ctx.sizes = tuple(demods), tuple(m is not None for m in mags), sizes
        return tuple(w_hats) + tuple(s_effs) + tuple(ds)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        lib = _lib.load()
        demods, sizes = ctx.demods, ctx.sizes
        nl, nd = len(demods), sum(demods)
        saved = ctx.saved_tensors
        flat, w_hats, styles, ds = saved[0], saved[1:1 + nd], saved[1 + nd:1 + nd + nl], saved[1 + nd + nl:1 + 2 * nd + nl]
        mag_it = iter(saved[1 + 2 * nd + nl:])
        mags = [next(mag_it) if h else None for h in ctx.has_mag]
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in gs]
        g_hats, g_ss, g_ds = gs[:nd], gs[nd:nd + nl], gs[nd + nl:]
        dev = styles[0].device
        n = int(styles[0].shape[0])
        need_w = [ctx.needs_input_grad[1 + k] for k in range(nd)]
        need_t = [ctx.needs_input_grad[1 + nd + l] for l in range(nl)]
        wsp_sizes = []
        k = 0
        for l in range(nl):
            o = int(w_hats[k].shape[0]) if demods[l] else 0
            wsp_sizes.append(int(lib.afcm_modulation_bank_workspace_floats(n, int(styles[l].shape[1]), o, int(demods[l]))))
            k += int(demods[l])
        wsp = torch.empty([sum(wsp_sizes)], dtype=torch.float32, device=dev)
        table = (_lib.ModulationLayer * nl)()
        dws, dts = [], []
        off, woff, k = 0, 0, 0
        for l in range(nl):
            L, t = table[l], styles[l]
            L.cin, L.demodulate = int(t.shape[1]), int(demods[l])
            L.t, L.magnitude = t.data_ptr(), _lib.ptr(mags[l])
            L.r = flat.data_ptr() + 4 * (sum(sizes) + l)
            L.g_s = _lib.ptr(g_ss[l])
            dt = torch.empty_like(t)
            dts.append(dt)
            L.dt = dt.data_ptr()
            L.workspace = wsp.data_ptr() + 4 * woff
            woff += wsp_sizes[l]
            if demods[l]:
                w_hat = w_hats[k]
                o, i, kh, kw = w_hat.shape
                L.cout, L.kk = int(o), int(kh * kw)
                L.w_hat, L.d = w_hat.data_ptr(), ds[k].data_ptr()
                L.wsq = flat.data_ptr() + 4 * off
                L.scale = flat.data_ptr() + 4 * (off + o * i)
                L.g_hat, L.g_d = _lib.ptr(g_hats[k]), _lib.ptr(g_ds[k])
                dw = torch.empty_like(w_hat) if need_w[k] else None
                dws.append(dw)
                L.dw = _lib.ptr(dw)
                off += sizes[k]
                k += 1
        _lib.check(lib.afcm_modulation_bank_bwd(table, nl, n, _lib.stream_ptr(styles[0])), 'modulation_bank_bwd')
        dts = [dt if nt else None for dt, nt in zip(dts, need_t)]
        return (None,) + tuple(dws) + tuple(dts) + (None,) * nl


def modulation_bank(items):
    """[(w_hat, in_scale, out_scale or None)] for a list of ``Item`` -- the tuples ``modulation_coefficients_fused`` returns, layer by
    layer.  A layer that does not demodulate gets its weight back (as fp32) and out_scale None."""
    _lib.require_gpu(*[it.weight for it in items], *[it.styles for it in items])
    demods = tuple(it.demodulate for it in items)
    nl, nd = len(items), sum(demods)
    out = _ModulationBank.apply(demods, *[it.weight for it in items if it.demodulate], *[it.styles for it in items],
                                *[it.magnitude for it in items])
    w_hats, s_effs, ds = iter(out[:nd]), out[nd:nd + nl], iter(out[nd + nl:])
    res = []
    for l, it in enumerate(items):
        if it.demodulate:
            res.append((next(w_hats), s_effs[l], next(ds)))
        else:
            res.append((it.weight.to(torch.float32), s_effs[l], None))
    return res
