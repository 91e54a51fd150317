"""Fold ensembling + segmentation export (oracle; test infrastructure only).

Restates reference e2enet/inference/predict.py:282-301 (sum of the per-fold float32 softmax volumes, ``/= len(params)``,
``transpose([0] + [i + 1 for i in transpose_backward])``) and e2enet/inference/segmentation_export.py:118-136 (argmax or
region thresholds, placement into the uint8 volume of the original size through ``crop_bbox``) in numpy, and the resampling
branch (:84-104) through scipy (``resample_softmax``: parity unpinned, see there); without the SimpleITK writer (:144-148).
"""
import numpy as np
from scipy import ndimage


RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3


def resample_softmax(data, new_shape, lowres_axis=None, order=1, order_z=0):
    """resample_data_or_seg(data, new_shape, is_seg=False, axis=[lowres_axis], order, do_separate_z=lowres_axis is not None,
    order_z) of reference e2enet/preprocessing/preprocessing.py:113-202, with its third-party call
    ``skimage.transform.resize(img, shape, order, mode='edge', anti_aliasing=False)`` (scikit-image 0.19.3, requirements.txt)
    restated as what that release evaluates: ``scipy.ndimage.zoom(img, shape / img.shape, order=order, mode='nearest',
    grid_mode=True)`` on the float64 image.  PARITY UNPINNED for this function: scikit-image is not in the image, so no golden
    could be produced from the reference; resize's final clip to the input's value range is a no-op for order <= 1 up to
    rounding and is not restated."""
    assert data.ndim == 4 and len(new_shape) == 3
    dtype = data.dtype
    shape = np.array(data[0].shape)
    new_shape = np.array(new_shape)
    if not np.any(shape != new_shape):
        return data
    data = data.astype(float)

    def resize(img, shp):
        return ndimage.zoom(img, np.array(shp, dtype=float) / np.array(img.shape), order=order, mode='nearest', grid_mode=True)
    if lowres_axis is None:
        return np.vstack([resize(data[c], new_shape)[None].astype(dtype) for c in range(data.shape[0])]).astype(dtype)
    axis = int(lowres_axis)
    new_shape_2d = np.delete(new_shape, axis)
    out = []
    for c in range(data.shape[0]):
        slices = [resize(np.take(data[c], i, axis=axis), new_shape_2d).astype(dtype) for i in range(shape[axis])]
        vol = np.stack(slices, axis)
        if shape[axis] != new_shape[axis]:
            rows, cols, dim = new_shape
            orig = vol.shape
            mr, mc, md = np.mgrid[:rows, :cols, :dim]
            coords = np.array([float(orig[0]) / rows * (mr + 0.5) - 0.5, float(orig[1]) / cols * (mc + 0.5) - 0.5,
                               float(orig[2]) / dim * (md + 0.5) - 0.5])
            out.append(ndimage.map_coordinates(vol, coords, order=order_z, mode='nearest')[None].astype(dtype))
        else:
            out.append(vol[None].astype(dtype))
    return np.vstack(out).astype(dtype)


def ensemble_softmax(softmaxes):
    total = softmaxes[0].copy()
    for s in softmaxes[1:]:
        total += s
    if len(softmaxes) > 1:
        total /= len(softmaxes)
    return total


def export_segmentation(softmax, properties_dict, transpose_backward=None, region_class_order=None, lowres_axis=None):
    if transpose_backward is not None:
        softmax = softmax.transpose([0] + [i + 1 for i in transpose_backward])
    after = properties_dict.get('size_after_cropping')
    if after is not None and any(int(a) != int(b) for a, b in zip(softmax.shape[1:], after)):
        softmax = resample_softmax(softmax, after, lowres_axis)          # segmentation_export.py:84-104
    if region_class_order is None:
        seg = softmax.argmax(0)
    else:
        seg = np.zeros(softmax.shape[1:])
        for i, c in enumerate(region_class_order):
            seg[softmax[i] > 0.5] = c
    bbox = properties_dict.get('crop_bbox')
    if bbox is not None:
        shape = properties_dict.get('original_size_of_raw_data')
        out = np.zeros(shape, dtype=np.uint8)
        bbox = [list(b) for b in bbox]
        for c in range(3):
            bbox[c][1] = np.min((bbox[c][0] + seg.shape[c], shape[c]))
        out[bbox[0][0]:bbox[0][1], bbox[1][0]:bbox[1][1], bbox[2][0]:bbox[2][1]] = \
            seg[:bbox[0][1] - bbox[0][0], :bbox[1][1] - bbox[1][0], :bbox[2][1] - bbox[2][0]]
        return out
    return seg.astype(np.uint8)
