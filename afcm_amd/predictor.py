"""Sliding-window volume prediction with halo removal (models/predictor.py:17-51,106-218).

The reference walks a loader of overlapping patches, runs ``model.test()`` (the EMA generator under ``no_grad``) on each
batch, strips ``patch_halo`` voxels from every patch side that is not on the volume border, accumulates the rest into a
[C, D, H, W] map and divides by the visit count.  This module keeps that arithmetic; the NIfTI writer (SimpleITK, absent
here) and the H5 dataset are out of scope -- the predictor returns the averaged arrays.
"""
import numpy as np
import torch


def remove_halo(patch, index, shape, patch_halo):
    """models/predictor.py:17-51.  patch [C, d, h, w]; index = (channel slice, z, y, x slices into the volume); shape = (D, H, W).
    Sides on the volume border keep their voxels, the others lose ``patch_halo`` voxels.  Returns (cropped patch, volume index)."""
    assert len(patch_halo) == 3

    def new_slices(slicing, max_size, pad):
        if slicing.start == 0:
            p_start, i_start = 0, 0
        else:
            p_start, i_start = pad, slicing.start + pad
        if slicing.stop == max_size:
            p_stop, i_stop = None, max_size
        else:
            # the reference's quirk, kept: a zero halo on an interior side yields patch[..., :1] (models/predictor.py:34)
            p_stop, i_stop = (-pad if pad != 0 else 1), slicing.stop - pad
        return slice(p_start, p_stop), slice(i_start, i_stop)

    D, H, W = shape
    i_c, i_z, i_y, i_x = index
    p_z, i_z = new_slices(i_z, D, patch_halo[0])
    p_y, i_y = new_slices(i_y, H, patch_halo[1])
    p_x, i_x = new_slices(i_x, W, patch_halo[2])
    return patch[(slice(0, patch.shape[0]), p_z, p_y, p_x)], (i_c, i_z, i_y, i_x)


def validate_halo(patch_halo, patch_shape, stride_shape):
    """models/predictor.py:227-237: neighbouring patches must overlap by at least the halo."""
    overlap = np.subtract(patch_shape, stride_shape)
    assert np.all(overlap - patch_halo >= 0), f'Not enough patch overlap for stride: {stride_shape} and halo: {patch_halo}'


def patch_indices(volume_shape, patch_shape, stride_shape):
    """Patch positions of the reference's slice builder (data/utils.py:38-124 restated: regular strides, the last patch of an
    axis pulled back to end at the border)."""
    def starts(n, k, s):
        assert n >= k, 'sample size has to be bigger than the patch size'
        out = list(range(0, n - k + 1, s))
        if out[-1] + k < n:
            out.append(n - k)
        return out
    D, H, W = volume_shape
    kd, kh, kw = patch_shape
    sd, sh, sw = stride_shape
    return [(slice(z, z + kd), slice(y, y + kh), slice(x, x + kw))
            for z in starts(D, kd, sd) for y in starts(H, kh, sh) for x in starts(W, kw, sw)]


class SlidingWindowPredictor:
    """``predict(volume_shape, batches)`` with ``batches`` yielding (prediction [B, C, d, h, w] tensor or array, indices) --
    the accumulation half of StandardPredictor.__call__ (models/predictor.py:158-199).  ``run(model_fn, volume, ...)`` is the
    whole loop for an in-memory [C_in, D, H, W] volume: cut patches, call ``model_fn(batch) -> [B, C, d, h, w]``, average."""

    def __init__(self, out_channels=1, patch_halo=(4, 8, 8), prediction_channel=None):
        self.out_channels, self.patch_halo, self.prediction_channel = int(out_channels), tuple(patch_halo), prediction_channel

    def allocate(self, volume_shape):
        shape = ((self.out_channels if self.prediction_channel is None else 1),) + tuple(volume_shape)
        return np.zeros(shape, dtype='float32'), np.zeros(shape, dtype='uint8')      # models/predictor.py:201-207

    def accumulate(self, prediction_map, normalization_mask, prediction, indices, volume_shape):
        prediction = prediction.detach().cpu().numpy() if isinstance(prediction, torch.Tensor) else np.asarray(prediction)
        for pred, index in zip(prediction, indices):
            channel_slice = slice(0, self.out_channels) if self.prediction_channel is None else slice(0, 1)
            index = (channel_slice,) + tuple(index)
            if self.prediction_channel is not None:
                pred = np.expand_dims(pred[self.prediction_channel], axis=0)
            u_prediction, u_index = remove_halo(pred, index, volume_shape, self.patch_halo)
            prediction_map[u_index] += u_prediction
            normalization_mask[u_index] += 1

    def predict(self, volume_shape, batches):
        prediction_map, normalization_mask = self.allocate(volume_shape)
        for prediction, indices in batches:
            self.accumulate(prediction_map, normalization_mask, prediction, indices, volume_shape)
        return prediction_map / normalization_mask                                    # models/predictor.py:215

    @torch.no_grad()
    def run(self, model_fn, volume, patch_shape, stride_shape, batch_size=1):
        volume_shape = tuple(volume.shape[1:])
        validate_halo(self.patch_halo, patch_shape, stride_shape)
        idx = patch_indices(volume_shape, patch_shape, stride_shape)

        def batches():
            for i in range(0, len(idx), batch_size):
                chunk = idx[i:i + batch_size]
                batch = torch.stack([torch.as_tensor(volume[(slice(None),) + ix]) for ix in chunk])
                yield model_fn(batch), chunk
        return self.predict(volume_shape, batches())
