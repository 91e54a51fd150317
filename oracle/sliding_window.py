"""Sliding-window inference (oracle; test infrastructure only).

Restates reference e2enet/network_architecture/neural_network.py:
  * _compute_steps_for_sliding_window   :260-284
  * _get_gaussian                       :244-258
  * _internal_maybe_mirror_and_pred_3D  :500-565
  * _internal_predict_3D_3Dconv_tiled   :286-426 (all_in_gpu=False branch, fp32)
and ``pad_nd_image`` of batchgenerators==0.24 (requirements.txt:1; symmetric
diff//2 / diff//2 + diff%2 constant padding plus the slicer that undoes it).
"""
from typing import Callable, Sequence
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def compute_steps(patch_size, image_size, step_size):
    target = [p * step_size for p in patch_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
    steps = []
    for dim in range(len(patch_size)):
        max_step = image_size[dim] - patch_size[dim]
        actual = max_step / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def gaussian_map(patch_size, sigma_scale=1. / 8):
    tmp = np.zeros(patch_size)
    tmp[tuple(i // 2 for i in patch_size)] = 1
    g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
    g = (g / np.max(g) * 1).astype(np.float32)
    g[g == 0] = np.min(g[g != 0])
    return g


def pad_to_patch(x: np.ndarray, patch_size, constant=0):
    """x: [C, X, Y, Z] -> (padded, slicer) with slicer over all 4 axes."""
    old = np.array(x.shape[-len(patch_size):])
    new = np.array([max(patch_size[i], old[i]) for i in range(len(patch_size))])
    diff = new - old
    below = diff // 2
    above = diff // 2 + diff % 2
    pad_list = [[0, 0]] * (x.ndim - len(patch_size)) + [list(i) for i in zip(below, above)]
    res = np.pad(x, pad_list, 'constant', constant_values=constant) if diff.any() else x
    pl = np.array(pad_list)
    pl[:, 1] = np.array(res.shape) - pl[:, 1]
    return res, [slice(*i) for i in pl]


_FLIPS = [(), (4,), (3,), (4, 3), (2,), (4, 2), (3, 2), (4, 3, 2)]       # order of :529-560


def mirror_predict(net_fn: Callable, x: torch.Tensor, num_classes, mirror_axes=(0, 1, 2),
                   do_mirroring=True, mult=None):
    """net_fn: [1,C,X,Y,Z] -> softmax probabilities [1,K,X,Y,Z]."""
    result = torch.zeros([1, num_classes] + list(x.shape[2:]), dtype=torch.float)
    n_flips = 8 if do_mirroring else 1
    num_results = 2 ** len(mirror_axes) if do_mirroring else 1
    for m in range(n_flips):
        dims = _FLIPS[m]
        if not all((d - 2) in mirror_axes for d in dims):
            continue
        if dims:
            pred = net_fn(torch.flip(x, dims))
            result += 1 / num_results * torch.flip(pred, dims)
        else:
            result += 1 / num_results * net_fn(x)
    if mult is not None:
        result[:, :] *= mult
    return result


def predict_tiled(net_fn: Callable, x: np.ndarray, num_classes, patch_size, step_size=0.5,
                  do_mirroring=True, mirror_axes=(0, 1, 2), use_gaussian=True, tile_filter=None):
    """Returns (seg int64 [X,Y,Z], probs float32 [K,X,Y,Z]).  ``tile_filter(idx)`` (optional)
    restricts which tiles are evaluated (used to model rank sharding)."""
    data, slicer = pad_to_patch(x, patch_size)
    steps = compute_steps(patch_size, data.shape[1:], step_size)
    num_tiles = len(steps[0]) * len(steps[1]) * len(steps[2])
    if use_gaussian and num_tiles > 1:
        g = gaussian_map(patch_size)
        add = g
        g_t = torch.from_numpy(g)
    else:
        g_t = None
        add = np.ones(patch_size, dtype=np.float32)
    agg = np.zeros([num_classes] + list(data.shape[1:]), dtype=np.float32)
    cnt = np.zeros([num_classes] + list(data.shape[1:]), dtype=np.float32)
    t = 0
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                if tile_filter is None or tile_filter(t):
                    tile = torch.from_numpy(np.ascontiguousarray(
                        data[None, :, sx:sx + patch_size[0], sy:sy + patch_size[1], sz:sz + patch_size[2]]))
                    pred = mirror_predict(net_fn, tile, num_classes, mirror_axes, do_mirroring, g_t)[0].numpy()
                    agg[:, sx:sx + patch_size[0], sy:sy + patch_size[1], sz:sz + patch_size[2]] += pred
                    cnt[:, sx:sx + patch_size[0], sy:sy + patch_size[1], sz:sz + patch_size[2]] += add
                t += 1
    sl = tuple([slice(0, agg.shape[0])] + slicer[1:])
    agg = agg[sl]
    cnt = cnt[sl]
    agg = agg / cnt
    return agg.argmax(0), agg
