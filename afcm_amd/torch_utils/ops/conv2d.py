"""Dense stride-1 convolution with per-plane input/output scaling on the MFMA kernels of
afcm_amd/csrc/conv2d.hip -- the compute behind ``modulated_conv2d`` (NET:25-64) and the encoder's
``conv2d_gradfix.conv2d`` call (NET:505).

    y[n, o] = out_scale[n, o] * sum_{i, r, s} w[o, i, r, s] * (in_scale[n, i] * x[n, i, p + r - pad, q + s - pad])

The weights are shared by the batch; the style modulation and the demodulation of the reference's
per-sample weights are the two per-plane scales (identical algebra).  Backward runs on
the same kernels: the data gradient is the forward kernel with transposed/flipped weights, the
weight gradient is a pixel-contraction GEMM, the scale gradients are per-plane dot products.
"""
import torch

from ... import _lib
from ... import profiling
from . import _rows


def _pad64(n):
    return (n + 63) // 64 * 64


def pack_weights(w, dtype, mode):
    """fp32 [O, I, k, k] -> the kernels' K-chunked layout in `dtype` (C ABI afcm_conv2d_pack_weights)."""
    lib = _lib.load()
    o, i, kh, kw = w.shape
    assert kh == kw and kh in (1, 3), 'only 1x1 and 3x3 kernels are supported'
    w = w.detach().to(torch.float32).contiguous()
    code = _lib._DTYPES[dtype]
    bk = lib.afcm_conv2d_block_k_ks(code, kh)
    rows, cols = (o, i) if mode == 0 else (i, o)
    rows_pad = _pad64(rows)
    nkc = (cols + bk - 1) // bk
    dst = torch.empty([nkc, kh * kw, rows_pad, bk], dtype=dtype, device=w.device)
    _lib.check(lib.afcm_conv2d_pack_weights(dst.data_ptr(), w.data_ptr(), code, o, i, kh, mode, rows_pad, _lib.stream_ptr(w)),
               'conv2d_pack_weights')
    return dst, rows_pad


def pack_weights_both(w, dtype):
    """Forward AND data-gradient images of one layer's weights in one launch (C ABI afcm_conv2d_pack_weights2):
    ((packed, rows_pad), (packed_t, rows_pad_t)) -- what pack_weights(w, dtype, 0) and pack_weights(w, dtype, 1) return."""
    lib = _lib.load()
    o, i, kh, kw = w.shape
    assert kh == kw and kh in (1, 3), 'only 1x1 and 3x3 kernels are supported'
    w = w.detach().to(torch.float32).contiguous()
    code = _lib._DTYPES[dtype]
    bk = lib.afcm_conv2d_block_k_ks(code, kh)
    rp0, rp1 = _pad64(o), _pad64(i)
    d0 = torch.empty([(i + bk - 1) // bk, kh * kw, rp0, bk], dtype=dtype, device=w.device)
    d1 = torch.empty([(o + bk - 1) // bk, kh * kw, rp1, bk], dtype=dtype, device=w.device)
    _lib.check(lib.afcm_conv2d_pack_weights2(d0.data_ptr(), d1.data_ptr(), w.data_ptr(), code, o, i, kh, rp0, rp1, _lib.stream_ptr(w)),
               'conv2d_pack_weights2')
    return (d0, rp0), (d1, rp1)


PACK_MAX = _lib.PACK_MAX


def pack_weights_bank(ws, dtype, need_dgrad=True):
    """``pack_weights_both`` for a list of same-kernel-size weights in ONE launch (C ABI afcm_conv2d_pack_bank): a list of
    ((packed, rows_pad), (packed_t, rows_pad_t)), bit-identical to the per-layer calls.  ``need_dgrad=False`` (no gradient will be
    taken: inference, torch.no_grad()) builds the forward images only: (packed_t, rows_pad_t) is None."""
    lib = _lib.load()
    assert 0 < len(ws) <= _lib.PACK_MAX
    ks = int(ws[0].shape[2])
    assert ks in (1, 3) and all(tuple(w.shape[2:]) == (ks, ks) for w in ws), 'one kernel size (1x1 or 3x3) per bank'
    code = _lib._DTYPES[dtype]
    bk = lib.afcm_conv2d_block_k_ks(code, ks)
    table = (_lib.PackEntry * len(ws))()
    out, keep = [], []
    for e, w in zip(table, ws):
        o, i = int(w.shape[0]), int(w.shape[1])
        w32 = w.detach().to(torch.float32).contiguous()
        keep.append(w32)
        rp0, rp1 = _pad64(o), _pad64(i)
        d0 = torch.empty([(i + bk - 1) // bk, ks * ks, rp0, bk], dtype=dtype, device=w.device)
        d1 = torch.empty([(o + bk - 1) // bk, ks * ks, rp1, bk], dtype=dtype, device=w.device) if need_dgrad else None
        e.dst_fwd, e.dst_dgrad, e.w = d0.data_ptr(), (d1.data_ptr() if need_dgrad else 0), w32.data_ptr()
        e.cout, e.cin, e.rows_pad_fwd, e.rows_pad_dgrad = o, i, rp0, rp1
        out.append(((d0, rp0), ((d1, rp1) if need_dgrad else None)))
    _lib.check(lib.afcm_conv2d_pack_bank(table, len(ws), code, ks, _lib.stream_ptr(ws[0])), 'conv2d_pack_bank')
    return out


def scale_planes(x, scale, out_dtype=None):
    """y[n, c] = x[n, c] * scale[n, c] (scale None: cast only)."""
    lib = _lib.load()
    out_dtype = out_dtype or x.dtype
    full = _rows.whole_buffer(x) if out_dtype == x.dtype else None
    if full is not None:
        # a row-pitched x: one pass over its whole buffer (the padding rides along: elementwise, nothing depends on it), y comes
        # back with the same pitch
        return scale_planes(full, scale, out_dtype)[..., :x.shape[3]]
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    n, c, h, w = x.shape
    if scale is not None:
        scale = scale.to(torch.float32).contiguous()
        assert scale.numel() == n * c
    _lib.check(lib.afcm_scale_planes(y.data_ptr(), x.data_ptr(), _lib.ptr(scale), _lib.dtype_code(x), _lib._DTYPES[out_dtype],
                                     n * c, h * w, _lib.stream_ptr(x)), 'scale_planes')
    return y


def plane_dot(a, b=None):
    """[N, C] fp32: sum over H, W of a * b (b None: plain sum).  Row-pitched operands (_rows.py) are read in place."""
    lib = _lib.load()
    if b is not None:
        assert b.shape == a.shape and b.dtype == a.dtype
    lda = _rows.pitch_of(a)
    ldb = _rows.pitch_of(b) if b is not None else 0
    # the pitched kernel's preconditions (afcm_plane_dot_ld): both operands dense or row-pitched AS THEY ARE (pitch_of also checks
    # the 4-byte base alignment and an even pitch), an even width for 2-byte elements, rows of at least 16 bytes.  Anything else --
    # an expanded stride-0 gradient, an odd-width plane, a view at an odd element offset -- takes the dense kernel on a copy.
    if (not (a.is_contiguous() and (b is None or b.is_contiguous())) and lda is not None and ldb is not None
            and a.shape[3] * a.element_size() >= 16 and (a.element_size() == 4 or a.shape[3] % 2 == 0)
            and a.data_ptr() % 4 == 0 and (b is None or b.data_ptr() % 4 == 0)):      # (pitch_of takes a dense tensor at any offset)
        n, c, h, w = a.shape
        out = torch.empty([n, c], dtype=torch.float32, device=a.device)
        _lib.check(lib.afcm_plane_dot_ld(out.data_ptr(), a.data_ptr(), _lib.ptr(b), _lib.dtype_code(a), n * c, h, w, lda, ldb, _lib.stream_ptr(a)),
                   'plane_dot')
        return out
    a = a.contiguous()
    if a.data_ptr() % 16:                 # a view at an odd storage offset: the kernel wants 16-byte aligned bases
        a = a.clone()
    n, c, h, w = a.shape
    if b is not None:
        b = b.contiguous()
        assert b.shape == a.shape and b.dtype == a.dtype
        if b.data_ptr() % 16:
            b = b.clone()
    out = torch.empty([n, c], dtype=torch.float32, device=a.device)
    _lib.check(lib.afcm_plane_dot(out.data_ptr(), a.data_ptr(), _lib.ptr(b), _lib.dtype_code(a), n * c, h * w, _lib.stream_ptr(a)),
               'plane_dot')
    return out


def plane_dot_gated(a, b, flags, out_scale, gz, next_scale=None, gskip=None):
    """[N, C] fp32: out_scale * (gz - next_scale * gskip) for the planes whose ``flags`` ([N, C, slots] int32, afcm_filtered_lrelu_args.
    clamp_flags) are all zero, the real sum over H, W of a * b for the others and for planes whose two sums cancel below 1/8 of their
    size (C ABI afcm_plane_dot_gated_ld).  Operands outside the row kernel's preconditions: the plain dot product."""
    lib = _lib.load()
    assert b.shape == a.shape and b.dtype == a.dtype
    lda, ldb = _rows.pitch_of(a), _rows.pitch_of(b)
    n, c, h, w = a.shape
    ok = (lda is not None and ldb is not None and w * a.element_size() >= 16 and (a.element_size() == 4 or w % 2 == 0)
          and a.data_ptr() % 4 == 0 and b.data_ptr() % 4 == 0 and flags.dtype == torch.int32 and flags.is_contiguous()
          and tuple(flags.shape[:2]) == (n, c))
    if not ok:
        return plane_dot(a, b)
    f32 = torch.float32
    for t in (out_scale, gz, next_scale, gskip):
        assert t is None or (t.dtype == f32 and t.is_contiguous() and t.numel() == n * c)
    out = torch.empty([n, c], dtype=f32, device=a.device)
    _lib.check(lib.afcm_plane_dot_gated_ld(out.data_ptr(), a.data_ptr(), b.data_ptr(), _lib.dtype_code(a), n * c, h, w, lda, ldb,
                                           flags.data_ptr(), int(flags.shape[2]), out_scale.data_ptr(), gz.data_ptr(), _lib.ptr(next_scale),
                                           _lib.ptr(gskip), _lib.stream_ptr(a)), 'plane_dot_gated')
    return out


# ---- fp32 3x3 convolutions on the 16-bit matrix pipe (split operands: include/afcm_hip.h, afcm_split16 / afcm_conv2d_split) --------
# (part dtype, forward terms, data-gradient terms, weight-gradient terms) of an fp32 3x3 conv; None: the native fp32 MFMA kernels
# (0.157 PFLOP/s peak against 2.5 for the 16-bit types).
#   float16, 3 terms of a two-way split, operands scaled by a power of two to just below 2^15: 22 significand bits -- measured equal to
#     an fp32 dot product of the same length (3e-7 of the output scale at K = 4608).  The default.
#   bfloat16 (no scaling, any range): 3 terms keep ~16 bits (4e-6), the 6 terms of order <= 2 of a three-way split all 24 (2e-7).
# Dynamic range of the float16 route -- ONE power-of-two bound per tensor (amax_bits over all N * C planes; a bound per
# sample would not factor out of the weight gradient's sum over samples):
#   * an element more than ~2^18 below the tensor's largest keeps only its first part (11 bits): a sample or plane that small next to
#     the rest of the batch is computed to float16, not fp32, precision;
#   * an inf / NaN anywhere sets the scale to 1; finite |v| > 65504 in that tensor then become inf in float16 -- more non-finites than
#     the native kernels would produce, for a tensor that already held one.
# Activations of a normalised generator sit within a few powers of two of each other, which is why this is the default; a caller with
# wider tensors sets FP32_SPLIT = (torch.bfloat16, 6, 6, 6) (any range, 24 bits) or None (the native fp32 kernels).
FP32_SPLIT = (torch.float16, 3, 3, 3)
_SPLIT_TERMS = {           # (part of the activations, part of the weights) per term, smallest products first
    3: ((1, 0), (0, 1), (0, 0)),
    6: ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)),
}


def _parts_fit(shape):
    """The parts of a [N, C, H, W] tensor (up to three, each the tensor's size rounded up to 4 elements, 2 bytes per element) lie inside
    the split kernel's 2^31-byte scalar offsets (afcm_conv2d_split: (parts - 1) * part stride + one image < 2^31 bytes)."""
    n, c, h, w = [int(v) for v in shape]
    numel = n * c * h * w
    return (numel + 3) // 4 * 4 * 2 * 2 + c * h * w * 2 < (1 << 31)


def _split_plan(x, ks, cout=None, padding=None):
    """The split-operand route takes fp32 tensors on the GPU, 3x3 kernels, even widths and parts inside the kernel's 2^31-byte
    scalar offsets -- of the input AND, when ``cout`` / ``padding`` say what the output is, of the gradient the backward will split
    (a conv whose forward fitted but whose dy [N, cout, P, Q] did not failed in backward with "part stride out of range";
    decided here it takes the native fp32 kernels in both directions instead)."""
    if FP32_SPLIT is None or x.dtype != torch.float32 or ks != 3 or x.device.type != 'cuda' or x.ndim != 4 or x.shape[3] % 2:
        return None
    if not _parts_fit(x.shape):
        return None
    if cout is not None and padding is not None:
        p, q = x.shape[2] + 2 * padding - 2, x.shape[3] + 2 * padding - 2
        if q % 2 or not _parts_fit((x.shape[0], cout, p, q)):
            return None
    return FP32_SPLIT


_BOUND_POOL = {}       # device -> [zeroed int32 words, next free index]


def _bound_word(device):
    """A zeroed int32 word on `device` (a view of a pooled zeros tensor: one fill launch per 1024 words)."""
    slot = _BOUND_POOL.get(device)
    if slot is None or slot[1] >= slot[0].numel():
        slot = _BOUND_POOL[device] = [torch.zeros([1024], dtype=torch.int32, device=device), 0]
    word = slot[0][slot[1]:slot[1] + 1]
    slot[1] += 1
    return word


def amax_bits(x, scale=None, out=None):
    """One int32 word on the device holding the bit pattern of max |scale[n, c] * x[n, c]| (C ABI afcm_amax_bits; ``out``: a word to
    raise instead of a fresh one).  split16 / pack_weights_split / the split conv turn it into the power of two g that brings that
    magnitude into [2^14, 2^15) -- float16 parts of g * v cannot overflow (max 65504) and the second part of everything above 2^-18 of
    the largest magnitude is a normal number.  One pass over x, no host round trip."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.is_contiguous()
    word = _bound_word(x.device) if out is None else out
    if x.ndim == 4:
        planes, hw = x.shape[0] * x.shape[1], x.shape[2] * x.shape[3]
    else:
        planes, hw = 1, x.numel()
    if scale is not None:
        scale = scale.detach().to(torch.float32).contiguous()
        assert scale.numel() == planes
    _lib.check(lib.afcm_amax_bits(word.data_ptr(), x.data_ptr(), planes, hw, _lib.ptr(scale), _lib.stream_ptr(x)), 'amax_bits')
    return word


def pow2_factor(word):
    """The factor the kernels derive from a bound word (host-side mirror, for tests): 2^(15 - e) with bound = f 2^e, f in [0.5, 1)."""
    b = word.view(torch.float32).double().cpu()
    if not bool(torch.isfinite(b).all()):
        return 1.0
    return float(2.0 ** (15 - torch.frexp(b.clamp_min(1e-30))[1].item()))


def split16(x, scale=None, nparts=2, dtype=torch.float16, bound=None):
    """[nparts, N, C, H, W] of `dtype`: g * scale[n, c] * x[n, c] as a sum of `nparts` 16-bit tensors (C ABI afcm_split16); every
    part is dense, ``.stride(0)`` is the part stride.  ``bound``: the word of ``amax_bits`` (g: its power of two) or None (g = 1)."""
    lib = _lib.load()
    x = x.contiguous()
    assert x.dtype == torch.float32 and x.ndim == 4 and nparts in (2, 3) and dtype in (torch.float16, torch.bfloat16)
    n, c, h, w = x.shape
    total = x.numel()
    stride = (total + 3) // 4 * 4
    buf = torch.empty([nparts, stride], dtype=dtype, device=x.device)
    if scale is not None:
        scale = scale.to(torch.float32).contiguous()
        assert scale.numel() == n * c
    if bound is not None:
        assert bound.dtype == torch.int32 and bound.numel() == 1 and bound.device == x.device
    _lib.check(lib.afcm_split16(buf.data_ptr(), x.data_ptr(), _lib.ptr(scale), _lib.ptr(bound), _lib._DTYPES[dtype], n * c, h * w, nparts, stride,
                                _lib.stream_ptr(x)), 'split16')
    return buf[:, :total].view(nparts, n, c, h, w)


def plane_dot_parts(parts, b, bound=None):
    """[N, C] fp32: sum over H, W of (sum of the 16-bit parts) * b / g (C ABI afcm_plane_dot_parts) -- plane_dot(x, b) for an x that was kept as
    its split16 parts (factor g from ``bound``)."""
    lib = _lib.load()
    nparts, n, c, h, w = parts.shape
    b = b.contiguous()
    assert b.dtype == torch.float32 and tuple(b.shape) == (n, c, h, w) and parts.stride(1) == c * h * w
    out = torch.empty([n, c], dtype=torch.float32, device=b.device)
    _lib.check(lib.afcm_plane_dot_parts(out.data_ptr(), parts.data_ptr(), parts.stride(0), nparts, b.data_ptr(), _lib._DTYPES[parts.dtype], n * c, h * w,
                                        _lib.ptr(bound), _lib.stream_ptr(b)), 'plane_dot_parts')
    return out


def _split_operand(x, scale, terms, dtype, bound=None):
    """(parts, bound): the parts of scale * x for a `terms`-term product; float16 parts carry the power of two of the magnitude word
    ``bound`` (given: the same tensor and scale were measured before)."""
    x = x.contiguous()
    if dtype == torch.float16 and bound is None:
        bound = amax_bits(x, scale)
    return split16(x, scale, _nparts(terms), dtype, bound if dtype == torch.float16 else None), (bound if dtype == torch.float16 else None)


def _nparts(*term_counts):
    return 1 + max(max(a, b) for t in term_counts for a, b in _SPLIT_TERMS[t])


def pack_weights_split(w, terms, dtype, transposed=False):
    """(packed, rows_pad, bound): the packed 16-bit image of the stacked weight parts of `w` ([O, I, 3, 3] fp32; ``transposed``: of the
    data gradient's [I, O, 3, 3] flipped kernel) for a `terms`-term split conv (C ABI afcm_conv2d_pack_split) -- channel block t holds
    the weight part of term t, zero-padded to a multiple of 16 channels; float16 parts are those of g * w, g from the magnitude word ``bound``."""
    lib = _lib.load()
    w = w.detach().to(torch.float32).contiguous()
    o, i = int(w.shape[0]), int(w.shape[1])
    assert tuple(w.shape[2:]) == (3, 3)
    rows, cols = (i, o) if transposed else (o, i)
    rows_pad = _pad64(rows)
    bound = amax_bits(w) if dtype == torch.float16 else None
    table = _SPLIT_TERMS[terms]
    bk = lib.afcm_conv2d_block_k_ks(_lib._DTYPES[dtype], 3)
    dst = torch.empty([terms * ((cols + bk - 1) // bk), 9, rows_pad, bk], dtype=dtype, device=w.device)
    code = sum(b << (4 * t) for t, (_, b) in enumerate(table))
    _lib.check(lib.afcm_conv2d_pack_split(dst.data_ptr(), w.data_ptr(), _lib.ptr(bound), _lib._DTYPES[dtype], o, i, 1 if transposed else 0, rows_pad,
                                          terms, code, _lib.stream_ptr(w)), 'conv2d_pack_split')
    return dst, rows_pad, bound


def _conv_split(parts, wp, rows_pad, terms, oscale, cout, pad, obias=None, bounds=(None, None)):
    """fp32 y = oscale * conv(w, x) + obias from the 16-bit parts of x (split16) and the stacked weight image (pack_weights_split);
    ``bounds``: the two operands' magnitude words, their factors undone in the kernel's epilogue."""
    lib = _lib.load()
    nparts, n, cin, h, w = parts.shape
    table = _SPLIT_TERMS[terms]
    assert nparts > max(a for a, _ in table) and parts.dtype == wp.dtype and parts.stride(1) == cin * h * w
    bk = lib.afcm_conv2d_block_k_ks(_lib._DTYPES[parts.dtype], 3)
    assert tuple(wp.shape) == (terms * ((cin + bk - 1) // bk), 9, rows_pad, bk), 'the weight image does not belong to this split'
    p, q = h + 2 * pad - 2, w + 2 * pad - 2
    y = torch.empty([n, cout, p, q], dtype=torch.float32, device=parts.device)
    if oscale is not None:
        oscale = oscale.to(torch.float32).contiguous()
        assert oscale.numel() == n * cout
    if obias is not None:
        obias = obias.to(torch.float32).contiguous()
        assert obias.numel() == cout
    code = sum(a << (4 * t) for t, (a, _) in enumerate(table))
    span = profiling.span('conv2d', 2.0 * n * cout * cin * 9 * p * q)
    _lib.check(lib.afcm_conv2d_split(y.data_ptr(), parts.data_ptr(), wp.data_ptr(), _lib.ptr(oscale), _lib.ptr(obias), _lib._DTYPES[parts.dtype], n, cin,
                                     cout, h, w, pad, rows_pad, terms, code, parts.stride(0), _lib.ptr(bounds[0]), _lib.ptr(bounds[1]),
                                     _lib.stream_ptr(parts)), 'conv2d_split')
    if span is not None:
        span.end()
    return y


def _wgrad_split(dy_parts, x_parts, cout, cin, pad, terms, bounds=(None, None)):
    """fp32 weight gradient of a 3x3 conv from the 16-bit parts of dy and x: one 16-bit weight-gradient launch per term, summed."""
    dw = None
    framed = {}            # a dy part is framed for the pad-1 route once, not once per term that uses it
    for a, b in _SPLIT_TERMS[terms]:
        if b not in framed:
            framed[b] = _frame_dy(dy_parts[b], 3, pad)
        dyb, padb = framed[b]
        d = _wgrad_raw(dyb, x_parts[a], cout, cin, 3, padb, work_share=1.0 / terms)
        dw = d if dw is None else dw.add_(d)
    if bounds[0] is not None or bounds[1] is not None:
        _lib.check(_lib.load().afcm_unscale(dw.data_ptr(), dw.numel(), _lib.ptr(bounds[0]), _lib.ptr(bounds[1]), _lib.stream_ptr(dw)), 'unscale')
    return dw


def _pitch_conv(dtype, ks):
    """Does the conv kernel for this case address rows by pitch?  (The 16-bit 3x3 kernel; C ABI afcm_conv2d_ld.)"""
    return dtype in (torch.bfloat16, torch.float16) and ks == 3


def _pitch_wgrad(dtype, ks, pad):
    """The 16-byte LDS-DMA weight-gradient kernel (C ABI afcm_conv2d_wgrad_ld)."""
    return dtype in (torch.bfloat16, torch.float16) and ((ks == 3 and pad == 2) or (ks == 1 and pad == 0))


def _conv_raw(x, wp, rows_pad, oscale, cout, ks, pad, obias=None, pitched_out=False):
    """x dense or row-pitched (_rows.py; kernels without pitch support get a contiguous copy); ``pitched_out``: y row-pitched
    where the kernel can write one."""
    lib = _lib.load()
    n, cin, h, w = x.shape
    p, q = h + 2 * pad - ks + 1, w + 2 * pad - ks + 1
    if _pitch_conv(x.dtype, ks):
        x, xld = _rows.rows(x)
        y = _rows.empty([n, cout, p, q], x.dtype, x.device, pitched=pitched_out)
    else:
        x, xld = _rows.dense(x), w
        y = torch.empty([n, cout, p, q], dtype=x.dtype, device=x.device)
    yld = q if y.is_contiguous() else y.stride(2)
    if oscale is not None:
        oscale = oscale.to(torch.float32).contiguous()
    span = profiling.span('conv2d', 2.0 * n * cout * cin * ks * ks * p * q)
    if obias is not None:
        obias = obias.to(torch.float32).contiguous()
        assert obias.numel() == cout
    _lib.check(lib.afcm_conv2d_ld(y.data_ptr(), x.data_ptr(), wp.data_ptr(), _lib.ptr(oscale), _lib.ptr(obias), _lib.dtype_code(x), n, cin, cout, h, w,
                                  ks, pad, rows_pad, 0 if xld == w else xld, 0 if yld == q else yld, _lib.stream_ptr(x)), 'conv2d')
    if span is not None:
        span.end()
    return y


# elements of dy up to which a pad-1 weight gradient is computed as pad-2 on a zero-framed dy (profiles/r05_wgrad_pad1.txt, batch 16:
# 32^2 x 512 = 8.4 M: 139 us framed against 207 on the dword kernel; 64^2 x 256 = 16.8 M: 150 against 123; larger planes: the dword kernel)
_FRAME_WGRAD_MAX = 12 << 20


def _frame_dy(dy, ks, pad):
    """(dy, pad) for the weight-gradient kernels: pad-1 weight gradients only have the dword LDS-DMA kernel (0.41 PF/s on the generator's
    512 -> 512 bottleneck conv at 36^2: 236 us); with dy framed by one ring of zeros the same sums are a pad-2 weight gradient, which the
    16-byte granule kernel takes.  Worth it while the framing copy is small (the discriminator's large planes: measured, no gain)."""
    if ks == 3 and pad == 1 and dy.dtype in (torch.bfloat16, torch.float16) and dy.numel() <= _FRAME_WGRAD_MAX:
        return torch.nn.functional.pad(dy, [1, 1, 1, 1]), 2
    return dy, pad


def _wgrad_raw(dy, x, cout, cin, ks, pad, work_share=1.0, dots_with=None):
    """``work_share``: the fraction of the algorithmic flops this launch stands for in the kernel timing (a split-operand term: 1 / terms).
    ``dots_with``: a weight tensor [cout, cin, ks, ks] -- return (dw, dots) with dots[n, i] = <x[n, i], conv^T(dots_with, dy)[n, i]> read from the
    per-image weight-gradient slabs (C ABI afcm_conv2d_wgrad_dots_ld), or (dw, None) where that form is not available."""
    lib = _lib.load()
    n, _, h, w = x.shape
    work = work_share * 2.0 * n * cout * cin * ks * ks * (h + 2 * pad - ks + 1) * (w + 2 * pad - ks + 1)      # algorithmic flops (before any framing)
    dy, pad = _frame_dy(dy, ks, pad)
    p = h + 2 * pad - ks + 1
    if _pitch_wgrad(x.dtype, ks, pad):
        (dy, lddy), (x, ldx) = _rows.rows(dy), _rows.rows(x)
    else:
        (dy, lddy), (x, ldx) = (_rows.dense(dy), dy.shape[3]), (_rows.dense(x), w)
    splits = lib.afcm_conv2d_wgrad_splits(n, cout, cin, p)
    dw = torch.empty([cout, cin, ks, ks], dtype=torch.float32, device=x.device)
    ws = torch.empty([splits, cout, cin, ks, ks], dtype=torch.float32, device=x.device)
    span = profiling.span('conv2d_wgrad', work)
    dots = None
    if dots_with is not None and WGRAD_DOTS and x.dtype in (torch.bfloat16, torch.float16):
        wref = dots_with.detach().to(torch.float32).contiguous()
        dots = torch.empty([n, cin], dtype=torch.float32, device=x.device)
        rc = _lib.check(lib.afcm_conv2d_wgrad_dots_ld(dw.data_ptr(), dots.data_ptr(), ws.data_ptr(), dy.data_ptr(), x.data_ptr(), wref.data_ptr(),
                                                      _lib.dtype_code(x), n, cin, cout, h, w, ks, pad, 0 if lddy == dy.shape[3] else lddy,
                                                      0 if ldx == w else ldx, _lib.stream_ptr(x)), 'conv2d_wgrad_dots')
        if rc != 0:
            dots = None
    if dots is None:
        _lib.check(lib.afcm_conv2d_wgrad_ld(dw.data_ptr(), ws.data_ptr(), dy.data_ptr(), x.data_ptr(), _lib.dtype_code(x), n, cin, cout,
                                            h, w, ks, pad, 0 if lddy == dy.shape[3] else lddy, 0 if ldx == w else ldx, _lib.stream_ptr(x)), 'conv2d_wgrad')
    if span is not None:
        span.end()
    return dw if dots_with is None else (dw, dots)


# the per-plane dot products <x, dx> from the weight gradient's per-image slabs where the split plan allows (module switch: tests compare)
WGRAD_DOTS = True


class _ScaledConv2d(torch.autograd.Function):
    """y = out_scale * conv(w, in_scale * x); scales are [N, C] fp32 tensors or None."""

    @staticmethod
    def forward(ctx, x, w, in_scale, out_scale, padding, prescaled=False, link=None):
        ctx.link = link if (link is not None and prescaled) else None
        _lib.require_gpu(x, w, in_scale, out_scale)
        if x.ndim != 4 or w.ndim != 4 or x.shape[1] != w.shape[1]:
            raise RuntimeError(f'conv2d: incompatible shapes x{tuple(x.shape)} w{tuple(w.shape)}')
        if x.numel() == 0:
            raise RuntimeError('x is empty')
        _lib.dtype_code(x)
        cout, cin, ks, _ = w.shape
        x = x.contiguous()
        # prescaled: the producer of x already applied in_scale (fused into its epilogue) and owns the gradient of in_scale
        # a backward that will need the data gradient gets its (transposed, flipped) weight image from the same launch
        ctx.wpt = None
        plan = _split_plan(x, ks, cout, padding)
        if plan is not None:
            # fp32 on the 16-bit matrix pipe.  The style factor is applied while splitting; the unscaled x is what the backward keeps
            dt, tf = plan[0], plan[1]
            eff_in = in_scale if (in_scale is not None and not prescaled) else None
            parts, gsx = _split_operand(x, eff_in, tf, dt)
            wp, rows_pad, gsw = pack_weights_split(w, tf, dt)
            y = _conv_split(parts, wp, rows_pad, tf, out_scale, cout, padding, bounds=(gsx, gsw))
            # A modulated conv (scales present: never differentiated twice) keeps the PARTS of s * x for its backward -- the weight
            # gradient's operand as it is, the style gradient through plane_dot_parts, same bytes as x -- and not x; a plain conv
            # (the discriminator's: R1 differentiates its backward) keeps x and splits it again.
            keep_parts = (in_scale is not None or out_scale is not None) and _nparts(plan[3]) <= parts.shape[0]
            ctx.save_for_backward(parts if keep_parts else x, w, in_scale, out_scale, y if (out_scale is not None and ctx.needs_input_grad[3]) else None)
            ctx.padding = padding
            ctx.prescaled = bool(prescaled)
            ctx.split = plan
            ctx.kept_parts = keep_parts
            ctx.x_bound = gsx                 # the magnitude word of s * x
            return y
        ctx.split = None
        xs = scale_planes(x, in_scale) if (in_scale is not None and not prescaled) else x
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            (wp, rows_pad), ctx.wpt = pack_weights_both(w, x.dtype)
        else:
            wp, rows_pad = pack_weights(w, x.dtype, 0)
        y = _conv_raw(xs, wp, rows_pad, out_scale, cout, ks, padding)
        ctx.save_for_backward(xs, w, in_scale, out_scale, y if (out_scale is not None and ctx.needs_input_grad[3]) else None)
        ctx.padding = padding
        ctx.prescaled = bool(prescaled)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, w, in_scale, out_scale, y = ctx.saved_tensors
        pad = ctx.padding
        if torch.is_grad_enabled() and (dy.requires_grad or xs.requires_grad or w.requires_grad):
            # a higher-order graph is being recorded (R1 penalty of the discriminator, models/comodgan_model.py:143-147): the
            # gradients are themselves convolutions, so express them with the same differentiable nodes
            if in_scale is not None or out_scale is not None:
                raise RuntimeError('conv2d: second-order gradients are only built for the unscaled convolution')
            ks = int(w.shape[2])
            dx = dw = None
            if ctx.needs_input_grad[0]:
                dx = _ScaledConv2d.apply(dy, w.transpose(0, 1).flip([2, 3]), None, None, ks - 1 - pad, False)
            if ctx.needs_input_grad[1]:
                dw = _ConvWgrad.apply(dy, xs, ks, pad).to(w.dtype)
            return dx, dw, None, None, None, None, None
        dy = dy.detach()
        if ctx.prescaled:
            in_scale = None
        cout, cin, ks, _ = w.shape
        dy = dy.contiguous()
        dx = dw = d_in = d_out = None
        if ctx.split is not None:
            # fp32 on the 16-bit matrix pipe: dy's parts (the demodulation factor applied while splitting) feed both gradients; here
            # xs is the UNSCALED x
            dt, td, tw = ctx.split[0], ctx.split[2], ctx.split[3]
            need_dx = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
            terms = [td] * bool(need_dx) + [tw] * bool(ctx.needs_input_grad[1])
            if terms:
                dparts, gsd = _split_operand(dy, out_scale, max(terms), dt)
            if need_dx:
                wpt, rows_pad, gsw = pack_weights_split(w, td, dt, transposed=True)
                dx = _conv_split(dparts, wpt, rows_pad, td, in_scale, cin, ks - 1 - pad, bounds=(gsd, gsw))
                if ctx.needs_input_grad[2] and in_scale is not None:
                    s1 = in_scale.to(torch.float32)
                    if ctx.kept_parts:
                        # d in_scale[n,i] = sum_pix x * g with the parts holding s * x and dx = s * g  =>  <s x, dx> / s^2
                        s2 = s1.square()
                        d_in = torch.where(s2 > 0, plane_dot_parts(xs, dx, ctx.x_bound) / s2.clamp_min(1e-30), torch.zeros_like(s2)).to(in_scale.dtype)
                    else:
                        # ... = <x, dx> / s
                        d_in = torch.where(s1 != 0, plane_dot(xs, dx) / torch.where(s1 != 0, s1, torch.ones_like(s1)), torch.zeros_like(s1)).to(in_scale.dtype)
            if ctx.needs_input_grad[1]:
                if ctx.kept_parts:
                    xparts, gsx = xs, ctx.x_bound
                else:
                    xparts, gsx = _split_operand(xs, in_scale, tw, dt, bound=ctx.x_bound)
                dw = _wgrad_split(dparts, xparts, cout, cin, pad, tw, bounds=(gsd, gsx)).to(w.dtype)
            if ctx.needs_input_grad[3]:
                d_out = (plane_dot(dy, y) / out_scale.to(torch.float32)).to(out_scale.dtype)
            return dx, dw, d_in, d_out, None, None, None
        dys = scale_planes(dy, out_scale) if out_scale is not None else dy
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            wpt, rows_pad = ctx.wpt if (ctx.wpt is not None and ctx.wpt[0].dtype == dy.dtype) else pack_weights(w, dy.dtype, 1)
            # data gradient of a pad-p correlation = correlation of dy with the flipped kernel at pad k-1-p;
            # the style factor of dx rides in the epilogue scale.
            dx = _conv_raw(dys, wpt, rows_pad, in_scale, cin, ks, ks - 1 - pad)
            if ctx.needs_input_grad[2] and in_scale is not None:
                # d in_scale[n,i] = sum_pix x * g with xs = s*x and dx = s*g  =>  <xs, dx> / s^2
                s2 = in_scale.to(torch.float32).square()
                d_in = torch.where(s2 > 0, plane_dot(xs, dx) / s2.clamp_min(1e-30), torch.zeros_like(s2)).to(in_scale.dtype)
        if ctx.needs_input_grad[1]:
            if ctx.link is not None and ctx.link.want:
                # the producer of the (prescaled) input wants <xs, dx> per plane: from this weight gradient's per-image slabs (fused_layer.LayerLink)
                dw, ctx.link.gz = _wgrad_raw(dys, xs, cout, cin, ks, pad, dots_with=w)
                dw = dw.to(w.dtype)
            else:
                dw = _wgrad_raw(dys, xs, cout, cin, ks, pad).to(w.dtype)
        if ctx.needs_input_grad[3]:
            # y = d * c  =>  d d[n,o] = <dy, c> = <dy, y> / d
            d_out = (plane_dot(dy, y) / out_scale.to(torch.float32)).to(out_scale.dtype)
        return dx, dw, d_in, d_out, None, None, None


class _ConvWgrad(torch.autograd.Function):
    """dw[o,i,r,s] = sum_{n,p,q} dy[n,o,p,q] x[n,i,p+r-pad,q+s-pad] as a differentiable node: its own gradients are a forward
    convolution of x with the incoming tensor (w.r.t. dy) and a data-gradient convolution of dy with it (w.r.t. x)."""

    @staticmethod
    def forward(ctx, dy, x, ks, pad):
        dy, x = dy.contiguous(), x.contiguous()
        cout, cin = int(dy.shape[1]), int(x.shape[1])
        ctx.save_for_backward(dy, x)
        ctx.cfg = (ks, pad)
        plan = _split_plan(dy, ks) if _split_plan(x, ks) is not None else None
        if plan is not None:
            (dparts, gsd), (xparts, gsx) = _split_operand(dy, None, plan[3], plan[0]), _split_operand(x, None, plan[3], plan[0])
            return _wgrad_split(dparts, xparts, cout, cin, pad, plan[3], bounds=(gsd, gsx))
        return _wgrad_raw(dy, x, cout, cin, ks, pad)

    @staticmethod
    def backward(ctx, g):
        dy, x = ctx.saved_tensors
        ks, pad = ctx.cfg
        g_dy = g_x = None
        if ctx.needs_input_grad[0]:
            g_dy = _ScaledConv2d.apply(x, g, None, None, pad, False)
        if ctx.needs_input_grad[1]:
            g_x = _ScaledConv2d.apply(dy, g.transpose(0, 1).flip([2, 3]), None, None, ks - 1 - pad, False)
        return g_dy, g_x, None, None


ZERO_STUFF_UPFIRDN = True      # module switch (A/B): the stride-2 backward's zero-stuffed dy through upfirdn2d(up=2, one tap)
_ONE_TAP = {}


def _one_tap(device):
    f = _ONE_TAP.get(device)
    if f is None:
        f = _ONE_TAP[device] = torch.ones([1, 1], dtype=torch.float32, device=device)
    return f


class _StridedConv2d(torch.autograd.Function):
    """y = conv(x, w, pad) at stride 2, 16-bit 3x3 (C ABI afcm_conv2d_stride2): the even rows / columns of the stride-1 result -- same
    operands, fp32 accumulation, within 1 ulp of the stride-1 route (whose kernel sums 32-channel chunks since r05, this one 16-channel
    ones) -- at a quarter of its MFMAs.  The gradients of a strided correlation are stride-1 convolutions with the zero-stuffed dy: the
    backward is written with the differentiable nodes above, so first- and higher-order graphs (the discriminator's R1 penalty) come
    out the same way they did when the stride was a slice of the stride-1 result."""

    @staticmethod
    def forward(ctx, x, w, padding):
        _lib.require_gpu(x, w)
        lib = _lib.load()
        n, cin, h, wd = x.shape
        cout = int(w.shape[0])
        x = x.contiguous()
        code = _lib._DTYPES[x.dtype]
        bk = lib.afcm_conv2d_block_k(code)
        rows_pad = (cout + 127) // 128 * 128                       # the stride-2 kernel runs 128-row blocks only
        # the image is packed per call (8 us): every caller in the package passes a temporary (`weight * weight_gain`, then `.to(dtype)`), so
        # the per-tensor cache r04 / r05 kept here never hit -- and a cache keyed on anything but the tensor object served
        # stale images
        w32 = w.detach().to(torch.float32).contiguous()
        wp = torch.empty([(cin + bk - 1) // bk, 9, rows_pad, bk], dtype=x.dtype, device=x.device)
        _lib.check(lib.afcm_conv2d_pack_weights_bk(wp.data_ptr(), w32.data_ptr(), code, cout, cin, 3, 0, rows_pad, bk, _lib.stream_ptr(x)), 'conv2d_pack_weights')
        p, q = (h + 2 * padding - 3) // 2 + 1, (wd + 2 * padding - 3) // 2 + 1
        y = torch.empty([n, cout, p, q], dtype=x.dtype, device=x.device)
        span = profiling.span('conv2d', 2.0 * n * cout * cin * 9 * p * q)
        _lib.check(lib.afcm_conv2d_stride2(y.data_ptr(), x.data_ptr(), wp.data_ptr(), code, n, cin, cout, h, wd, padding, rows_pad, _lib.stream_ptr(x)),
                   'conv2d_stride2')
        if span is not None:
            span.end()
        ctx.save_for_backward(x, w)
        ctx.padding = padding
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        pad = ctx.padding
        n, _, h, wd = x.shape
        fh, fw = h + 2 * pad - 2, wd + 2 * pad - 2
        if ZERO_STUFF_UPFIRDN and fw % 2 == 0 and dy.shape[3] % 2 == 0:
            # zero-stuffed dy = the gradient of the stride-1 result: upfirdn2d with up 2 and a one-tap filter writes it in ONE pass (row
            # kernel: 16-byte stores), differentiable like the slice assignment it replaces (a fill + a strided copy: two passes)
            from . import upfirdn2d as _upf
            full = _upf.upfirdn2d(dy, _one_tap(dy.device), up=2, padding=[0, fw - 2 * dy.shape[3], 0, fh - 2 * dy.shape[2]])
        else:
            full = dy.new_zeros([n, dy.shape[1], fh, fw])
            full[:, :, ::2, ::2] = dy                                # zero-stuffed: the gradient of the stride-1 result
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _ScaledConv2d.apply(full, w.to(torch.float32).transpose(0, 1).flip([2, 3]).contiguous(), None, None, 2 - pad, False)
        if ctx.needs_input_grad[1]:
            dw = _ConvWgrad.apply(full, x, 3, pad).to(w.dtype)
        return dx, dw, None


def strided_conv2d_supported(x, w, padding):
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and x.ndim == 4 and tuple(w.shape[2:]) == (3, 3) and x.shape[1] == w.shape[1]
            and x.shape[3] % 2 == 0 and 0 <= padding <= 2 and x.shape[1] * x.shape[2] * x.shape[3] * 2 < (1 << 31))


def strided_conv2d(x, w, padding=0):
    """3x3 correlation at stride 2 of 16-bit activations (see _StridedConv2d); check ``strided_conv2d_supported`` first."""
    return _StridedConv2d.apply(x.contiguous(), w.contiguous(), int(padding))


def scaled_conv2d(x, w, in_scale=None, out_scale=None, padding=0, prescaled=False, link_in=None):
    # contiguity is established out here, under autograd, so that the node saves tensors that are still part of the graph
    # (a second-order backward differentiates through them)
    return _ScaledConv2d.apply(x.contiguous(), w.contiguous(), in_scale, out_scale, int(padding), bool(prescaled), link_in)


def modulation_coefficients(w, s, demodulate=True, input_gain=None):
    """The small-tensor half of ``modulated_conv2d`` (NET:41-57): returns (w_hat [O,I,k,k], in_scale [N,I], out_scale [N,O]
    or None), all fp32 and differentiable, such that  y = out_scale * conv(w_hat, in_scale * x)."""
    n, i = int(s.shape[0]), int(w.shape[1])
    w = w.to(torch.float32)
    s = s.to(torch.float32)
    d = None
    if demodulate:
        w = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt()      # NET:42
        s = s * s.square().mean().rsqrt()                              # NET:43 (whole batch)
        # NET:50-52: rsqrt(sum_{i,k} (w[o,i,k] s[n,i])^2 + 1e-8), factorised as s^2 @ (sum_k w^2)^T
        d = (s.square() @ w.square().sum([2, 3]).t() + 1e-8).rsqrt()   # [N, O]
    if input_gain is not None:
        s = s * input_gain.to(torch.float32).expand(n, i)              # NET:55-57
    return w, s, d


class _WeightNorm(torch.autograd.Function):
    """(w_hat, wsq) = (w * rsqrt(mean w^2) per output channel, sum_k w_hat^2) -- C ABI afcm_weight_norm_fwd / _bwd."""

    @staticmethod
    def forward(ctx, w):
        lib = _lib.load()
        w = w.detach().to(torch.float32).contiguous()
        o, i, kh, kw = w.shape
        w_hat = torch.empty_like(w)
        wsq = torch.empty([o, i], dtype=torch.float32, device=w.device)
        scale = torch.empty([o], dtype=torch.float32, device=w.device)
        _lib.check(lib.afcm_weight_norm_fwd(w_hat.data_ptr(), wsq.data_ptr(), scale.data_ptr(), w.data_ptr(), o, i, kh * kw,
                                            _lib.stream_ptr(w)), 'weight_norm_fwd')
        ctx.save_for_backward(w_hat, scale)
        return w_hat, wsq

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_hat, g_wsq):
        w_hat, scale = ctx.saved_tensors
        lib = _lib.load()
        o, i, kh, kw = w_hat.shape
        g_hat = None if g_hat is None else g_hat.to(torch.float32).contiguous()
        g_wsq = None if g_wsq is None else g_wsq.to(torch.float32).contiguous()
        dw = torch.empty_like(w_hat)
        _lib.check(lib.afcm_weight_norm_bwd(dw.data_ptr(), _lib.ptr(g_hat), _lib.ptr(g_wsq), w_hat.data_ptr(), scale.data_ptr(), o, i, kh * kw,
                                            _lib.stream_ptr(w_hat)), 'weight_norm_bwd')
        return dw


class _StyleCoefs(torch.autograd.Function):
    """(s_eff, d) from the raw styles t [N, I], wsq [O, I] and the layer's magnitude EMA -- C ABI afcm_style_coefs_fwd / _bwd."""

    @staticmethod
    def forward(ctx, t, wsq, magnitude, demodulate):
        lib = _lib.load()
        t = t.detach().to(torch.float32).contiguous()
        n, i = t.shape
        o = int(wsq.shape[0]) if demodulate else 0
        s_eff = torch.empty_like(t)
        d = torch.empty([n, o], dtype=torch.float32, device=t.device) if demodulate else torch.empty([0], dtype=torch.float32, device=t.device)
        r = torch.empty([1], dtype=torch.float32, device=t.device)
        wsq_c = wsq.detach().contiguous() if demodulate else None
        mag = None if magnitude is None else magnitude.detach().to(torch.float32).reshape(1)
        _lib.check(lib.afcm_style_coefs_fwd(s_eff.data_ptr(), d.data_ptr() if demodulate else None, r.data_ptr(), t.data_ptr(), _lib.ptr(wsq_c),
                                            _lib.ptr(mag), n, i, o, int(bool(demodulate)), _lib.stream_ptr(t)), 'style_coefs_fwd')
        ctx.save_for_backward(t, wsq_c, mag, d, r)
        ctx.demodulate = bool(demodulate)
        if not demodulate:
            ctx.mark_non_differentiable(d)
        return s_eff, d

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_s, g_d):
        t, wsq, mag, d, r = ctx.saved_tensors
        lib = _lib.load()
        n, i = t.shape
        o = int(wsq.shape[0]) if ctx.demodulate else 0
        g_s = None if g_s is None else g_s.to(torch.float32).contiguous()
        g_d = None if (g_d is None or not ctx.demodulate) else g_d.to(torch.float32).contiguous()
        dt = torch.empty_like(t)
        g_wsq = torch.empty_like(wsq) if (ctx.demodulate and ctx.needs_input_grad[1]) else None
        ws = torch.empty([n * i + n * o + n * ((i + 63) // 64)], dtype=torch.float32, device=t.device)
        _lib.check(lib.afcm_style_coefs_bwd(dt.data_ptr(), _lib.ptr(g_wsq), ws.data_ptr(), _lib.ptr(g_s), _lib.ptr(g_d), t.data_ptr(),
                                            d.data_ptr() if ctx.demodulate else None, _lib.ptr(wsq), _lib.ptr(mag), r.data_ptr(), n, i, o,
                                            int(ctx.demodulate), _lib.stream_ptr(t)), 'style_coefs_bwd')
        return dt, g_wsq, None, None


def modulation_coefficients_fused(w, styles, demodulate=True, magnitude=None):
    """``modulation_coefficients`` with input_gain = magnitude.rsqrt() (the only form SynthesisLayer uses, NET:346) on the
    fused HIP kernels: 2 launches forward, 3 backward, instead of ~35 eager ones."""
    _lib.require_gpu(w, styles, magnitude)
    if demodulate:
        w_hat, wsq = _WeightNorm.apply(w)
        s_eff, d = _StyleCoefs.apply(styles, wsq, magnitude, True)
        return w_hat, s_eff, d
    s_eff, _ = _StyleCoefs.apply(styles, None, magnitude, False)
    return w.to(torch.float32), s_eff, None


def modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=None):
    """Drop-in for the reference's ``modulated_conv2d(x, w, s, demodulate, padding, input_gain)`` (NET:25-64).

    x [N, I, H, W]; w [O, I, k, k]; s [N, I]; input_gain [], [I] or [N, I].  Returns [N, O, H + 2p - k + 1, ...].
    """
    n = int(x.shape[0])
    o, i, kh, kw = w.shape
    assert x.shape[1] == i and tuple(s.shape) == (n, i)
    w, s, d = modulation_coefficients(w, s, demodulate, input_gain)
    if isinstance(padding, (list, tuple)):
        assert padding[0] == padding[1]
        padding = padding[0]
    return scaled_conv2d(x, w, s, d, padding)
