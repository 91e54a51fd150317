"""Fold ensembling + segmentation export (oracle; test infrastructure only).

Restates reference e2enet/inference/predict.py:282-301 (sum of the per-fold float32 softmax volumes, ``/= len(params)``,
``transpose([0] + [i + 1 for i in transpose_backward])``) and e2enet/inference/segmentation_export.py:118-136 (argmax or
region thresholds, placement into the uint8 volume of the original size through ``crop_bbox``) in numpy, without the
resampling branch (:73-104, third-party skimage) and the SimpleITK writer (:144-148).
"""
import numpy as np


def ensemble_softmax(softmaxes):
    total = softmaxes[0].copy()
    for s in softmaxes[1:]:
        total += s
    if len(softmaxes) > 1:
        total /= len(softmaxes)
    return total


def export_segmentation(softmax, properties_dict, transpose_backward=None, region_class_order=None):
    if transpose_backward is not None:
        softmax = softmax.transpose([0] + [i + 1 for i in transpose_backward])
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
