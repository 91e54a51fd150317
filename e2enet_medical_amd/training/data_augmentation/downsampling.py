"""Deep-supervision targets on the device (the loss-side end of the training input feed).

Mirrors reference e2enet/training/data_augmentation/downsampling.py:69-107 (``DownsampleSegForDSTransform2`` /
``downsample_seg_for_ds_transform2``): the label map of a batch is resized to every deep-supervision scale with
batchgenerators' ``resize_segmentation(seg, new_shape, order=0)`` = ``skimage.transform.resize(order=0, mode="edge",
anti_aliasing=False)``, which in the pinned scikit-image 0.19.3 is ``scipy.ndimage.zoom(order=0, mode="nearest",
grid_mode=True)``: ``out[j] = in[floor((j + 0.5) * in_size / out_size)]`` per axis, evaluated in float64.  In the
reference this runs inside the CPU augmentation workers; here the full-resolution labels are uploaded once and the
scales are gathered by a HIP kernel (``e2e_ds_target_gather``), so the host pipeline ships 1/1.14 of the bytes and the
trainer accepts a generator that yields the full-resolution target only.

Only ``order=0`` (what the trainer uses, data_augmentation_moreDA.py:199-204) is implemented; the rest of the
augmentation pipeline (batchgenerators spatial / intensity transforms) stays host tooling.
"""
from typing import Sequence

import numpy as np
import torch

from ..._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def zoom_nearest_indices(in_size: int, out_size: int) -> np.ndarray:
    """scipy.ndimage.zoom(order=0, mode='nearest', grid_mode=True) source index per output index, in the float64
    arithmetic of scipy's NI_ZoomShift: cc = (j + 0.5) * (in / out) - 0.5; index = floor(cc + 0.5), clamped."""
    zoom = float(in_size) / float(out_size)
    j = np.arange(out_size, dtype=np.float64)
    cc = (j + 0.5) * zoom - 0.5
    idx = np.floor(cc + 0.5).astype(np.int64)
    return np.clip(idx, 0, in_size - 1).astype(np.int32)


_INDEX_CACHE = {}


def _indices(in_size, out_size, device):
    key = (in_size, out_size, device.index)
    t = _INDEX_CACHE.get(key)
    if t is None:
        t = torch.from_numpy(zoom_nearest_indices(in_size, out_size)).to(device)
        _INDEX_CACHE[key] = t
    return t


def downsample_seg_for_ds_transform2(seg: torch.Tensor, ds_scales=((1, 1, 1), (0.5, 0.5, 0.5), (0.25, 0.25, 0.25)), order=0,
                                     axes=None):
    """seg: float GPU tensor [B, C, D, H, W].  Returns the list of targets, one per scale (the input itself for scale 1),
    shapes ``round(shape * scale)`` like the reference (:96-100)."""
    if order != 0:
        raise NotImplementedError("deep-supervision targets are resized with order 0 (nearest) only "
                                  "(data_augmentation_moreDA.py:199-204)")
    if not (isinstance(seg, torch.Tensor) and seg.is_cuda and seg.dtype == torch.float32 and seg.dim() == 5):
        raise RuntimeError("downsample_seg_for_ds_transform2 (MI355X engine) needs a float32 GPU tensor [B, C, D, H, W]")
    if axes is None:
        axes = list(range(2, seg.dim()))
    if list(axes) != [2, 3, 4]:
        raise NotImplementedError("axes other than the three spatial ones")
    seg = seg.contiguous()
    B, C, D, H, W = seg.shape
    out = []
    for s in ds_scales:
        if all(i == 1 for i in s):
            out.append(seg)
            continue
        new = np.array([D, H, W]).astype(float)
        for i in range(3):
            new[i] *= s[i]
        d, h, w = (int(v) for v in np.round(new).astype(int))
        o = torch.empty((B, C, d, h, w), dtype=torch.float32, device=seg.device)
        lib().ds_target_gather(seg.data_ptr(), o.data_ptr(), _indices(D, d, seg.device).data_ptr(),
                               _indices(H, h, seg.device).data_ptr(), _indices(W, w, seg.device).data_ptr(), B * C, D, H, W,
                               d, h, w, _stream())
        out.append(o)
    return out


class DownsampleSegForDSTransform2:
    """data_dict[output_key] = list of targets scaled by ds_scales (reference :69-84), on the device."""

    def __init__(self, ds_scales: Sequence = (1, 0.5, 0.25), order=0, input_key="seg", output_key="seg", axes=None):
        self.axes, self.output_key, self.input_key, self.order, self.ds_scales = axes, output_key, input_key, order, ds_scales

    def __call__(self, **data_dict):
        data_dict[self.output_key] = downsample_seg_for_ds_transform2(data_dict[self.input_key], self.ds_scales, self.order,
                                                                      self.axes)
        return data_dict
