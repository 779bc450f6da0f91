"""The slice dataset either side of the hot path (second half): registered MR volumes -> the tensors the training
step consumes, ``A [4, H, W]`` (four neighbouring thick slices of the input modality), ``B [1, H, W]`` (the target slice) and
``slice_idx`` (the target's fractional position between two thick slices -- the label ``c`` of G and D).

Restates, for the shipped `loaders` configuration (configs/adni/base.yml:40-62, cmsr.yml:17-22):
    data/cmsr_dataset.py:20-96    volumes of one subject, centre crop / constant pad to the patch size, slice positions
    data/cmsr_dataset.py:98-155   ``__getitem__``: thickness, modalities, the four thick slices around the target, slice_idx
    data/utils.py:38-124          SliceBuilder: patch origins along every axis (the last patch is pulled back inside the volume)
    data/augment/transforms.py:227-281 (CropToFixed, centred), :604-616 (Normalize to [-1, 1]), :647-663 (ToTensor)
The volumes come from any mapping ``{internal path: ndarray [D, H, W]}``; an ``.h5`` path is opened with h5py when that package
is installed (it is not in the build image, so that branch is untested here).  Parity: the slice positions are pinned through
``afcm_amd.predictor`` (same generator, golden P1); the rest of this module has no reference-held fixture and the reference's
dataset module does not import here (h5py, cv2, scikit-image) -- "parity unpinned", tested against closed forms
(tests/test_eval_predictor.py)."""
import random

import numpy as np
import torch

from .predictor import patch_indices


def crop_to_fixed(vol, size, mode='constant'):
    """Centre crop, then pad to ``size = (H, W)``; ``vol`` is [D, H, W] (transforms.py:250-275 with centered=True)."""
    assert vol.ndim == 3
    out = []
    pads = []
    for have, want in zip(vol.shape[1:], size):
        if want < have:
            out.append(((have - want) // 2, want))
            pads.append((0, 0))
        else:
            total = want - have
            out.append((0, want))
            pads.append((total // 2, total - total // 2))
    (y0, ny), (x0, nx) = out
    return np.pad(vol[:, y0:y0 + ny, x0:x0 + nx], ((0, 0), pads[0], pads[1]), mode=mode)


def normalize(m, min_value=0.0, max_value=255.0):
    """Min-max scaling to [-1, 1], clipped (transforms.py:609-616)."""
    assert max_value > min_value
    return np.clip(2 * ((m - min_value) / (max_value - min_value)) - 1, -1, 1)


def build_slices(shape, patch_shape, stride_shape):
    """Patch positions of a [D, H, W] volume as tuples of slices, z outermost (data/utils.py:93-124; the generator is
    afcm_amd.predictor.patch_indices, pinned to the reference's own by the P1 golden)."""
    return patch_indices(tuple(shape), tuple(patch_shape), tuple(stride_shape))


def open_volumes(source, internal_paths):
    """{path: ndarray [D, H, W]} from a mapping or an HDF5 file name (cmsr_dataset.py:84-96)."""
    if isinstance(source, str):
        try:
            import h5py
        except ImportError as e:
            raise RuntimeError(f'{source}: reading HDF5 volumes needs h5py, which is not installed; pass a mapping of arrays instead') from e
        with h5py.File(source, 'r') as f:
            source = {k: f[k][:] for k in internal_paths if k in f}
    vols = {}
    for k in internal_paths:
        assert k in source, f'Image {k} not found!'
        v = np.asarray(source[k])
        vols[k] = v[None] if v.ndim == 2 else v
    return vols


class SliceDataset(torch.utils.data.Dataset):
    """One subject's volumes as a map-style dataset of slices (AbstractHDF5Dataset, cmsr_dataset.py:20-158).

    ``raw_internal_path_in`` / ``raw_internal_path_out``: modalities of A / B; ``thickness``: candidate slice thicknesses (a random
    one per item when training, the first otherwise); ``slice_num``: 1 (the slice itself) or 4 (thick slices at -1, 0, +1, +2
    thicknesses around the target's own thick slice; positions outside the volume are zero planes BEFORE normalisation, i.e. they
    come out as the background value -1)."""

    def __init__(self, source, phase='train', patch_shape=(1, 256, 256), stride_shape=(1, 32, 32), raw_internal_path_in=('raw',),
                 raw_internal_path_out=('raw',), rand_output=False, cat_inputs=False, thickness=(), slice_num=4, min_value=0.0, max_value=255.0):
        assert phase in ('train', 'val', 'test')
        self.phase, self.rand_output, self.cat_inputs = phase, rand_output, cat_inputs
        self.raw_internal_path_in, self.raw_internal_path_out = list(raw_internal_path_in), list(raw_internal_path_out)
        self.thickness, self.slice_num = list(thickness), slice_num
        self.min_value, self.max_value = min_value, max_value
        paths = list(dict.fromkeys(self.raw_internal_path_in + self.raw_internal_path_out))
        self.raw = {k: crop_to_fixed(v, tuple(patch_shape[1:])) for k, v in open_volumes(source, paths).items()}
        self.raw_slices = build_slices(self.raw[self.raw_internal_path_out[-1]].shape, patch_shape, stride_shape)
        self.patch_count = len(self.raw_slices)

    def __len__(self):
        return self.patch_count

    def _plane(self, modality, where):
        """normalised float32 tensor [1, H, W] of one slice position (None: a zero plane, cmsr_dataset.py:138-139)."""
        v = self.raw[modality]
        m = v[where] if where is not None else np.zeros(v[0:1].shape)
        return torch.from_numpy(normalize(m, self.min_value, self.max_value).astype(np.float32))

    def __getitem__(self, idx):
        if idx >= len(self):
            raise StopIteration
        if len(self.thickness) > 0:
            thickness = random.choice(self.thickness) if self.phase == 'train' else self.thickness[0]
        else:
            thickness = -1
        if self.phase == 'train' and self.rand_output:
            modality_b = random.choice(self.raw_internal_path_out)
        else:
            modality_b = self.raw_internal_path_out[-1]
        modalities_a = [x for x in self.raw_internal_path_in if x != modality_b] if self.cat_inputs else [self.raw_internal_path_in[0]]
        planes = []
        idx_a = idx
        for modality in modalities_a:
            if self.slice_num == 1:
                planes.append(self._plane(modality, self.raw_slices[idx]))
            elif self.slice_num == 4:
                idx_a = int((idx // thickness) * thickness)                    # the thick slice the target falls into
                for pos in (idx_a - thickness, idx_a, idx_a + thickness, idx_a + 2 * thickness):
                    planes.append(self._plane(modality, self.raw_slices[pos] if 0 <= pos <= self.patch_count - 1 else None))
            else:
                raise NotImplementedError(f'slice number {self.slice_num} not supported')
        a = torch.cat(planes)
        slice_idx = np.array([idx - idx_a], dtype=np.float32) / thickness
        if self.phase == 'test':
            return a, torch.from_numpy(slice_idx), self.raw_slices[idx]
        onehot = np.zeros(len(self.raw_internal_path_out), dtype=np.float32)
        onehot[len(self.raw_internal_path_out) - 1] = 1
        return {'A': a, 'B': self._plane(modality_b, self.raw_slices[idx]), 'B_class': onehot, 'B_idx': torch.tensor([float(idx)]),
                'slice_idx': slice_idx}


def cmsr_dataset(sources, phase='train', **kwargs):
    """All subjects chained (CmsrDataset, cmsr_dataset.py:251-254): ``sources`` = mappings or HDF5 file names."""
    return torch.utils.data.ConcatDataset([SliceDataset(s, phase=phase, **kwargs) for s in sources])
