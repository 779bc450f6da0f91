"""Row-pitched activation tensors: the MI355X layout of the 16-bit activation stream between the kernels of a generator layer.

The generator's planes are 276, 278, 148, 150, 84, 86 ... elements wide: rows of 552, 556, 296 ... bytes, none of them a multiple
of the 128-byte memory line.  Every row segment a kernel stores then straddles two lines, and measured on the filtered_lrelu
kernels that is the difference between 2.7 and 4.2 TB/s (tools/bench_flrelu_align.py: same kernel, output 312 vs 320 wide).
So the tensors that the fused layer node produces (afcm_amd/torch_utils/ops/fused_layer.py) are allocated as [N, C, H, pitch] with
pitch = W rounded up to 64 bytes (``pitch_for``) and handed on as the strided view ``buf[..., :W]`` -- an ordinary torch tensor to everything
else (autograd, torch ops, ``.contiguous()``), while the HIP kernels that understand a pitch (C ABI ``*_pitch`` arguments) read
and write it in place.  Rules of the layout:

* columns >= W of a row are padding.  No kernel may depend on what they hold, except that a kernel that WRITES a pitched tensor
  fills them with finite values (the weight gradient multiplies the head of dy's padding by zeros, include/afcm_hip.h).
* a tensor that is neither dense nor of exactly this form is made contiguous (``rows``).
"""
import torch

# Module attributes, not environment switches (r05: the package reads ONE environment variable, AFCM_HIP_LIB in _lib.py, which selects an
# experimental build of the library; the measurements these three were set for are in docs/history_r01-r04.md) -- tools set them directly.
ROW_BYTES = 64          # rows padded to multiples of this (>= 16)
MAX_OVERHEAD = 0.10     # ... unless that pads a row by more than this fraction
ENABLED = True          # False = dense tensors everywhere


def pitch_for(w, dtype):
    """Row pitch (elements) of a ``w``-wide plane: the next multiple of 64 bytes -- measured on the filtered_lrelu forward kernels
    over eight generator layers: dense 1.85 ms, rows on 16-byte boundaries 1.74, on 64-byte boundaries 1.46, on 128-byte lines
    1.51 -- unless that pads the row by more than 10 % (the 84- / 52- / 36-wide planes: the extra bytes cost what the alignment
    wins), in which case the tensor stays dense."""
    e = max(ROW_BYTES, 16) // torch.empty([], dtype=dtype).element_size()
    ld = (w + e - 1) // e * e
    return ld if ld <= w * (1.0 + MAX_OVERHEAD) else w


def empty(shape, dtype, device, pitched=True):
    """An uninitialised [N, C, H, W] tensor; 16-bit dtypes with ``pitched``: rows on 64-byte boundaries (a view of [N, C, H, pitch])."""
    n, c, h, w = shape
    if pitched and ENABLED and dtype in (torch.bfloat16, torch.float16):
        ld = pitch_for(w, dtype)
        if ld != w:
            return torch.empty([n, c, h, ld], dtype=dtype, device=device)[..., :w]
    return torch.empty([n, c, h, w], dtype=dtype, device=device)


def pitch_of(t):
    """Row pitch (elements) if ``t`` is dense or a row-pitched view as made by ``empty``; None for any other layout."""
    if t.ndim != 4:
        return None
    if t.is_contiguous():
        return t.shape[3]
    n, c, h, w = t.shape
    s = t.stride()
    ld = s[2]
    if s[3] != 1 or ld < w or s[1] != h * ld or s[0] != c * h * ld or (ld & 1) or (t.data_ptr() & 3):
        return None
    return ld


def rows(t):
    """(tensor, row pitch): ``t`` itself when a pitch-aware kernel can address it, else a contiguous copy."""
    ld = pitch_of(t)
    if ld is None:
        t = t.contiguous()
        ld = t.shape[3]
    return t, ld


def dense(t):
    """For kernels that take dense tensors only."""
    return t if t.is_contiguous() else t.contiguous()


def whole_buffer(t):
    """The dense [N, C, H, pitch] tensor behind a row-pitched view (for elementwise kernels: the padding rides along), or None
    when ``t`` is dense, not of that form, or its storage does not cover the last row's padding."""
    if t.ndim != 4 or t.is_contiguous():
        return None
    ld = pitch_of(t)
    if ld is None:
        return None
    n, c, h, w = t.shape
    if (t.storage_offset() + n * c * h * ld) * t.element_size() > t.untyped_storage().nbytes():
        return None
    return torch.as_strided(t, (n, c, h, ld), (c * h * ld, h * ld, ld, 1))
