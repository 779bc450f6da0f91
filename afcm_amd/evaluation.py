"""Image-quality metrics of the reference's validation loop: util/evaluation.py on numpy arrays.

The reference delegates to scikit-image (``skimage.metrics.peak_signal_noise_ratio`` / ``structural_similarity``,
util/evaluation.py:3-4), which the reference pins to scikit-image 0.19.3 (requirements.txt:3) and which is absent from this
image; the two functions are restated here from their published definitions (Wang et al. 2004 for SSIM with scikit-image's
defaults: 7-wide uniform window, K1 = 0.01, K2 = 0.03, sample covariance, border of (win - 1) // 2 cropped before the mean).
PARITY UNPINNED against scikit-image itself (the package cannot be imported here and the reference holds no metric fixtures) --
tests pin the restatement to brute-force window loops and closed forms only.  Known numerical difference: scikit-image 0.19
keeps float32 images in float32 (``_supported_float_type``) where this restatement computes in float64 -- ~1e-6 dB of PSNR on
the reference's float32 inputs, far below the 0.05 dB bound the north star states.

Callers (train.py:86-99): images mapped to [0, 1] by (x + 1) / 2 and clipped, then ``evaluate_2D``.
"""
import numpy as np
from scipy import ndimage


def _float_data_range(img_true, signed_is_two=True):
    """scikit-image's default for floating-point images (dtype range (-1, 1)): 1 for a non-negative reference image, else 2."""
    lo, hi = float(np.min(img_true)), float(np.max(img_true))
    if hi > 1 or lo < -1:
        raise ValueError('image_true has intensity values outside the range expected for its data type; specify data_range')
    return 1.0 if lo >= 0 else 2.0


def peak_signal_noise_ratio(image_true, image_test, data_range=None):
    """10 log10(data_range^2 / MSE) in float64 (skimage.metrics.peak_signal_noise_ratio)."""
    a, b = np.asarray(image_true, dtype=np.float64), np.asarray(image_test, dtype=np.float64)
    if a.shape != b.shape:
        raise ValueError('input images must have the same dimensions')
    if data_range is None:
        data_range = _float_data_range(a)
    err = np.mean((a - b) ** 2)
    with np.errstate(divide='ignore'):
        return float(10 * np.log10((data_range ** 2) / err))


def structural_similarity(im1, im2, win_size=7, data_range=None, K1=0.01, K2=0.03, use_sample_covariance=True):
    """Mean SSIM over the image interior, any number of dimensions (2-D slices and 3-D volumes in the reference).
    ``data_range`` None follows the scikit-image releases the reference was written against: the span of the float dtype
    range, 2 (newer releases make the argument mandatory for float images)."""
    x, y = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    if x.shape != y.shape:
        raise ValueError('input images must have the same dimensions')
    if win_size % 2 != 1:
        raise ValueError('window size must be odd')
    if any(s < win_size for s in x.shape):
        raise ValueError('win_size exceeds image extent')
    if data_range is None:
        data_range = 2.0
    npix = win_size ** x.ndim
    cov_norm = npix / (npix - 1) if use_sample_covariance else 1.0
    filt = lambda v: ndimage.uniform_filter(v, size=win_size)
    ux, uy = filt(x), filt(y)
    uxx, uyy, uxy = filt(x * x), filt(y * y), filt(x * y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    c1, c2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win_size - 1) // 2
    return float(s[tuple(slice(pad, n - pad) for n in s.shape)].mean())


compare_psnr, compare_ssim = peak_signal_noise_ratio, structural_similarity       # the reference's import aliases (:3-4)


def psnr_2D(Gimg, Limg):
    """util/evaluation.py:31-37: each image divided by its own maximum first."""
    l, g = np.squeeze(Limg), np.squeeze(Gimg)
    return compare_psnr(l / l.max(), g / g.max())


def evaluate_2D(Gimg, Limg):
    """util/evaluation.py:92-104 for batches [N, 1, 1, H, W] (train.py:88-97): mean PSNR / SSIM over the slices whose target is
    not empty, and the reference's "mse" term (the mean absolute error of the WHOLE batch, added once per counted slice).
    Returns None when every target slice is empty."""
    c_psnr = c_ssim = c_mse = 0.0
    count = 0
    for i in range(Gimg.shape[0]):
        if np.max(Limg[i]) <= 0:
            continue
        c_psnr += psnr_2D(Gimg[i][0], Limg[i][0])
        c_ssim += compare_ssim(np.squeeze(Limg[i][0]), np.squeeze(Gimg[i][0]))
        c_mse += np.mean(np.abs(Limg - Gimg))
        count += 1
    if count == 0:
        return None
    return c_psnr / count, c_ssim / count, c_mse / count


def ThreeD_slice_psnr(Gimg, Limg):
    """:71-80: axial slices with a non-empty target, max-normalised."""
    c, count = 0.0, 0
    for i in range(Limg.shape[0]):
        if np.max(Limg[i]) <= 0:
            continue
        l, g = np.squeeze(Limg[i]), np.squeeze(Gimg[i])
        c += compare_psnr(l / l.max(), g / g.max())
        count += 1
    return c / count


def ThreeD_slice_ssim(Gimg, Limg):
    """:21-28."""
    c, count = 0.0, 0
    for i in range(Limg.shape[0]):
        if np.max(Limg[i]) <= 0:
            continue
        c += compare_ssim(Limg[i], Gimg[i])
        count += 1
    return c / count


def ThreeD_psnr(Gimg, Limg):
    """:40-68: PSNR of every slice along all three axes with the pair's joint value range as data_range; a constant pair adds
    the running mean (the reference's handling of empty slices)."""
    c = 0.0
    seen = 0
    for axis in range(3):
        for i in range(Gimg.shape[axis]):
            l, g = np.squeeze(np.take(Limg, i, axis=axis)), np.squeeze(np.take(Gimg, i, axis=axis))
            d_range = np.max([l, g]) - np.min([l, g])
            if d_range == 0:
                c += c / (seen + i + 1)
            else:
                c += compare_psnr(l, g, data_range=d_range)
        seen += Gimg.shape[axis]
    return c / sum(Gimg.shape)


def ThreeD_ssim(Gimg, Limg):
    """:6-18."""
    c = 0.0
    for axis in range(3):
        for i in range(Gimg.shape[axis]):
            c += compare_ssim(np.squeeze(np.take(Limg, i, axis=axis)), np.squeeze(np.take(Gimg, i, axis=axis)))
    return c / sum(Gimg.shape)


def evaluate_one(Gimg, Limg):
    """:107-114: (3-axis PSNR, 3-axis SSIM, MAE) of one volume."""
    return ThreeD_psnr(Gimg, Limg), ThreeD_ssim(Gimg, Limg), float(np.mean(np.abs(Limg - Gimg)))


def evaluate_slice(Gimg, Limg):
    """:116-121."""
    return ThreeD_slice_psnr(Gimg, Limg), ThreeD_slice_ssim(Gimg, Limg), float(np.mean(np.abs(Limg - Gimg)))


def evaluate_3D(Gimg, Limg):
    """:123-127: volume-level PSNR / SSIM (7^3 window)."""
    return compare_psnr(Limg, Gimg), compare_ssim(Limg, Gimg), float(np.mean(np.abs(Limg - Gimg)))


def to_unit_range(x):
    """train.py:93-96: [-1, 1] network range -> [0, 1], clipped."""
    return np.clip((np.asarray(x, dtype=np.float32) + 1) / 2, 0, 1)
