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

from . import upfirdn2d


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
    """conv2d_resample.py:29-53 without the cuDNN channels-last workaround (contiguous NCHW only here)."""
    if not flip_weight:            # F.conv2d is a correlation (flip_weight=True); flip for a true convolution
        w = w.flip([2, 3])
    op = F.conv_transpose2d if transpose else F.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
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
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)
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
