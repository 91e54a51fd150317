"""Deep-supervision target downsampling (oracle; test infrastructure only).

Restates reference e2enet/training/data_augmentation/downsampling.py:87-107 with order 0.  PARITY UNPINNED for this
file: the reference resizes with batchgenerators 0.24 ``resize_segmentation`` -> scikit-image 0.19.3
``skimage.transform.resize(order=0, mode="edge", anti_aliasing=False)`` (requirements.txt:1, :44), and neither package
is in this image, so no golden could be generated from the reference.  scikit-image 0.19.3 delegates that call to
``scipy.ndimage.zoom(input, new/old, order=0, mode="nearest", grid_mode=True)``; this oracle calls scipy's zoom itself
(scipy is here), i.e. it is anchored on the third-party routine the reference ends up in, not on reference outputs.
"""
import numpy as np
from scipy import ndimage


def downsample_seg_for_ds(seg: np.ndarray, ds_scales, order=0):
    assert order == 0
    out = []
    for s in ds_scales:
        if all(i == 1 for i in s):
            out.append(seg)
            continue
        new_shape = np.array(seg.shape).astype(float)
        for i, a in enumerate(range(2, seg.ndim)):
            new_shape[a] *= s[i]
        new_shape = np.round(new_shape).astype(int)
        o = np.zeros(new_shape, dtype=seg.dtype)
        for b in range(seg.shape[0]):
            for c in range(seg.shape[1]):
                src = seg[b, c].astype(float)
                zf = [n / o_ for n, o_ in zip(new_shape[2:], src.shape)]
                o[b, c] = ndimage.zoom(src, zf, order=0, mode="nearest", grid_mode=True).astype(seg.dtype)
        out.append(o)
    return out
