"""Training-time augmentation transforms, given their drawn parameters (oracle; test infrastructure only).

numpy / scipy restatement of the batchgenerators 0.24 transforms that reference
e2enet/training/data_augmentation/data_augmentation_moreDA.py:66-121 composes (third party, absent from the image and from
/root/reference: PARITY UNPINNED -- no golden could be generated; each function states the batchgenerators routine it follows).
Data interpolation orders are parameters (the reference: order_data = 3 in SpatialTransform, order_upsample = 3 in
SimulateLowResolution; order 1 is the device's other mode): scipy.ndimage itself is the independent implementation the kernels are
checked against.
"""
import numpy as np
from scipy import ndimage


def coordinate_mesh(patch_size, mat):
    """augment_spatial: create_zero_centered_coordinate_mesh, rotate_coords_3d / scale_coords (folded into A), + centre (t)."""
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) - (s - 1) / 2. for s in patch_size], indexing="ij")
    c = np.stack([g.reshape(-1) for g in grids])
    m = np.asarray(mat, dtype=np.float64).reshape(3, 4)
    return (m[:, :3] @ c + m[:, 3:4]).reshape((3,) + tuple(patch_size))


def spatial(data, seg, mats, patch_size, order_seg=1, cval_seg=-1.0, order_data=3):
    """SpatialTransform per sample: data map_coordinates(order=order_data, mode='constant', cval=0) [batchgenerators
    interpolate_img: img.astype(float) in, .astype(img.dtype) out; the reference passes order_data=3]; seg either
    order 0 with cval, or interpolate_img(is_seg=True, order=1): per label (ascending) a linear interpolation of the binary mask
    with the same cval, assigned where >= 0.5."""
    B = data.shape[0]
    out = np.zeros((B, data.shape[1]) + tuple(patch_size), np.float32)
    oseg = None if seg is None else np.zeros((B, seg.shape[1]) + tuple(patch_size), np.float32)
    for b in range(B):
        coords = coordinate_mesh(patch_size, mats[b])
        for c in range(data.shape[1]):
            out[b, c] = ndimage.map_coordinates(data[b, c].astype(float), coords, order=order_data, mode='constant', cval=0.0).astype(np.float32)
        if seg is not None:
            for c in range(seg.shape[1]):
                img = seg[b, c]
                if order_seg == 0:
                    oseg[b, c] = ndimage.map_coordinates(img.astype(float), coords, order=0, mode='constant', cval=cval_seg).astype(np.float32)
                else:
                    res = np.zeros(tuple(patch_size), np.float32)
                    for lab in np.unique(img):
                        m = ndimage.map_coordinates((img == lab).astype(float), coords, order=1, mode='constant', cval=cval_seg)
                        res[m >= 0.5] = lab
                    oseg[b, c] = res
    return out, oseg


def spatial_dummy_2d(data, seg, mats, patch_size, order_seg=1, cval_seg=-1.0, order_data=3):
    """The dummy_2D form (data_augmentation_moreDA.py:58-60, :80-81): Convert3DTo2DTransform reshapes [B, C, D, H, W] to
    [B, C * D, H, W], SpatialTransform interpolates every such image in 2-D with the sample's ONE in-plane affine (`mats` rows in
    the augmenter's 3 x 4 layout, slice row and column zero), Convert2DTo3DTransform reshapes back.  patch_size: (D, H, W)."""
    B, C, D = data.shape[:3]
    ps2 = tuple(patch_size[1:])
    out = np.zeros((B, C, D) + ps2, np.float32)
    oseg = None if seg is None else np.zeros((B, seg.shape[1], D) + ps2, np.float32)
    for b in range(B):
        m = np.asarray(mats[b], dtype=np.float64).reshape(3, 4)
        grids = np.meshgrid(*[np.arange(s, dtype=np.float64) - (s - 1) / 2. for s in ps2], indexing="ij")
        c = np.stack([g.reshape(-1) for g in grids])
        coords = (m[1:, 1:3] @ c + m[1:, 3:4]).reshape((2,) + ps2)
        for ch in range(C):
            for z in range(D):
                out[b, ch, z] = ndimage.map_coordinates(data[b, ch, z].astype(float), coords, order=order_data, mode='constant', cval=0.0).astype(np.float32)
        if seg is not None:
            for ch in range(seg.shape[1]):
                for z in range(D):
                    img = seg[b, ch, z]
                    if order_seg == 0:
                        oseg[b, ch, z] = ndimage.map_coordinates(img.astype(float), coords, order=0, mode='constant', cval=cval_seg).astype(np.float32)
                    else:
                        res = np.zeros(ps2, np.float32)
                        for lab in np.unique(img):
                            mm = ndimage.map_coordinates((img == lab).astype(float), coords, order=1, mode='constant', cval=cval_seg)
                            res[mm >= 0.5] = lab
                        oseg[b, ch, z] = res
    return out, oseg


def gaussian_blur(x, sigma):
    """augment_gaussian_blur: scipy.ndimage.gaussian_filter(channel, sigma, order=0)"""
    return ndimage.gaussian_filter(x.astype(np.float64), sigma, order=0).astype(np.float32)


def contrast(x, factor):
    """augment_contrast(preserve_range=True): (x - mean) * factor + mean, clipped to the channel's former [min, max]"""
    x = x.astype(np.float64)
    mn, lo, hi = x.mean(), x.min(), x.max()
    return np.clip((x - mn) * factor + mn, lo, hi).astype(np.float32)


def gamma(x, g, invert, retain_stats=True, eps=1e-7):
    """augment_gamma(per_channel=True) on one channel"""
    x = x.astype(np.float64)
    if invert:
        x = -x
    mn, sd = x.mean(), x.std()
    lo = x.min()
    rng = x.max() - lo
    x = np.power((x - lo) / float(rng + eps), g) * rng + lo
    if retain_stats:
        x = x - x.mean()
        x = x / (x.std() + 1e-8) * sd
        x = x + mn
    if invert:
        x = -x
    return x.astype(np.float32)


def _resize(img, shape, order):
    """skimage.transform.resize(order, mode='edge', anti_aliasing=False) of scikit-image 0.19.3 = scipy zoom, grid_mode"""
    return ndimage.zoom(img, np.array(shape, dtype=float) / np.array(img.shape), order=order, mode='nearest', grid_mode=True)


def low_resolution(x, zoom, order_upsample=3, ignore_axes=None):
    """augment_linear_downsampling_scipy on one channel: nearest down to round(shape * zoom), then up-sampling back with
    skimage resize(order_upsample, mode='edge', anti_aliasing=False) [the reference passes order_upsample=3]; resize's default
    clip=True clamps the result to the range of its input (a no-op for order <= 1); ignore_axes keep their size (dummy_2D: (0,))"""
    shp = np.array(x.shape)
    target = np.round(shp * zoom).astype(int)
    if ignore_axes is not None:
        for ax in ignore_axes:
            target[ax] = shp[ax]
    down = _resize(x.astype(np.float32), target, 0)
    up = _resize(down, shp, order_upsample)
    return np.clip(up, down.min(), down.max()).astype(np.float32)


def mirror(data, seg, axes_flags):
    """augment_mirroring on one sample: axes_flags[i] flips spatial axis i"""
    for ax in range(3):
        if axes_flags[ax]:
            data = np.flip(data, 1 + ax)
            seg = None if seg is None else np.flip(seg, 1 + ax)
    return np.ascontiguousarray(data), None if seg is None else np.ascontiguousarray(seg)


def finish(data, seg, use_mask):
    """MaskTransform(mask_idx_in_seg=0, set_outside_to=0) + RemoveLabelTransform(-1, 0)"""
    data, seg = data.copy(), seg.copy()
    if use_mask is not None:
        for b in range(data.shape[0]):
            m = seg[b, 0] < 0
            for c in range(data.shape[1]):
                if use_mask[c]:
                    data[b, c][m] = 0
    seg[seg == -1] = 0
    return data, seg
