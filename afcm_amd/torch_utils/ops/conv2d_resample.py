"""conv2d_resample: 2-D convolution with optional up/downsampling, the entry point the discriminator's (and the bottleneck's)
``Conv2dLayer`` calls -- ``conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True,
flip_filter=False)`` (models/networks/CoModGAN/torch_utils/ops/conv2d_resample.py:57-155).

Same decomposition as the reference: the resampling FIRs are ``upfirdn2d`` (here: the HIP kernel, arbitrarily
differentiable), the contraction is the framework's convolution exactly where the reference calls cuDNN through
``conv2d_gradfix`` (a plain ``F.conv2d`` on torch >= 1.11, conv2d_gradfix.py:53-55) -- on ROCm that is MIOpen.  The
discriminator needs stride-2 convolutions and a double backward (R1, models/comodgan_model.py:143-147), neither of which the
generator's MFMA conv kernels provide; padding is applied once at the beginning, not between the operations.
"""

import torch
import torch.nn.functional as F

from . import conv2d as _conv
from . import upfirdn2d

USE_MFMA_CONV = True      # module switch: False routes every convolution to the framework (A/B and debugging)
NATIVE_STRIDE2 = True     # (module attribute, no environment switch)  16-bit 3x3 stride-2 convolutions on the stride-2 MFMA kernel (False: the stride-1 result decimated, the r01/r02 route)
MFMA_CONV_FP32 = False    # fp32 activations too: correct, but the stride-by-decimation waste makes it slower than the framework's
                          # fp32 convolution over the whole D update (151.7 vs 148 ms, batch 16); the 16-bit blocks gain 2x (69 vs 139 ms)


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = [int(v) for v in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return padding


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d_resample.py:29-53 without the cuDNN channels-last workaround (contiguous NCHW only here).

    1x1 / 3x3, ungrouped, non-transposed convolutions of ROCm tensors run on the MFMA kernels of csrc/conv2d.hip (second-order
    differentiable through _ScaledConv2d / _ConvWgrad); a stride is taken by decimating the stride-1 result -- the windows of a
    strided correlation are a subset of the stride-1 ones.  Everything else (transposed, grouped, other kernel sizes, CPU) is the
    framework convolution, as in the reference."""
    if not flip_weight:            # F.conv2d is a correlation (flip_weight=True); flip for a true convolution
        w = w.flip([2, 3])
    pad = padding if isinstance(padding, int) else (padding[0] if padding[0] == padding[1] else None)
    kh, kw = int(w.shape[2]), int(w.shape[3])
    if (USE_MFMA_CONV and not transpose and groups == 1 and x.device.type == 'cuda' and kh == kw and kh in (1, 3) and pad is not None
            and 0 <= pad <= kh - 1 and x.dtype in ((torch.float32, torch.bfloat16, torch.float16) if MFMA_CONV_FP32 else (torch.bfloat16, torch.float16))
            and (x.dtype == torch.float32 or (x.shape[3] % 2 == 0 and (x.shape[3] + 2 * pad - kh + 1) % 2 == 0))):
        if stride == 2 and NATIVE_STRIDE2 and _conv.strided_conv2d_supported(x, w, pad):
            return _conv.strided_conv2d(x, w.to(torch.float32), pad)     # csrc/conv2d.hip conv2d_fwd16s2_kernel: no full-resolution intermediate
        y = _conv.scaled_conv2d(x, w.to(torch.float32), None, None, pad)
        if stride != 1:
            y = y[:, :, ::stride, ::stride]
        return y
    op = F.conv_transpose2d if transpose else F.conv2d
    return op(x, w.to(x.dtype), stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    # fp32 master weights are accepted next to 16-bit activations: the MFMA path packs them to the activation dtype itself (the same
    # single rounding as the reference's w.to(x.dtype)), the framework path casts at the call -- no conversion pass each way per layer
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and (w.dtype == x.dtype or w.dtype == torch.float32)
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1 and isinstance(groups, int) and groups >= 1
    out_channels, in_channels_per_group, kh, kw = [int(v) for v in w.shape]
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # padding adjusted for the resampling filters (conv2d_resample.py:96-106)
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2
        px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2
        py1 += (fh - down) // 2

    if kw == 1 and kh == 1 and (down > 1 and up == 1):            # 1x1 + down: downsample first (:109-112)
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if kw == 1 and kh == 1 and (up > 1 and down == 1):            # 1x1 + up: convolve first (:115-118)
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    if down > 1 and up == 1:                                      # down only: blur, strided conv (:121-124)
        # one extra padded column / row when the blurred size would be odd: no stride-`down` window reaches it, and the 16-bit
        # conv kernels want even widths
        ex = (x.shape[3] + px0 + px1 - (fw - 1)) & 1
        ey = (x.shape[2] + py0 + py1 - (fh - 1)) & 1
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1 + ex, py0, py1 + ey], flip_filter=flip_filter)
        y = _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)
        oh = (x.shape[2] - ey - kh) // down + 1
        ow = (x.shape[3] - ex - kw) // down + 1
        return y[:, :, :oh, :ow]
    if up > 1:                                                    # up (+ down): transposed strided conv (:127-143)
        if groups == 1:
            w = w.transpose(0, 1)
        else:
            w = w.reshape(groups, out_channels // groups, in_channels_per_group, kh, kw).transpose(1, 2)
            w = w.reshape(groups * in_channels_per_group, out_channels // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=w, stride=up, padding=[pyt, pxt], groups=groups, transpose=True, flip_weight=(not flip_weight))
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:       # plain conv (:146-148)
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)
    x = upfirdn2d.upfirdn2d(x=x, f=None, up=1, padding=[px0, px1, py0, py1], flip_filter=flip_filter)   # generic (:151-155)
    return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
