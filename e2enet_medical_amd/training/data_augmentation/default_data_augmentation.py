"""Default augmentation parameters and the loader patch size (reference
e2enet/training/data_augmentation/default_data_augmentation.py:39-131).  Values are the reference's; ``get_patch_size`` restates
batchgenerators' ``rotate_coords_3d`` (third party, 0.24) for the three single-axis rotations it needs."""
import os
from copy import deepcopy

import numpy as np

default_3D_augmentation_params = {
    "selected_data_channels": None,
    "selected_seg_channels": None,

    "do_elastic": True,
    "elastic_deform_alpha": (0., 900.),
    "elastic_deform_sigma": (9., 13.),
    "p_eldef": 0.2,

    "do_scaling": True,
    "scale_range": (0.85, 1.25),
    "independent_scale_factor_for_each_axis": False,
    "p_independent_scale_per_axis": 1,
    "p_scale": 0.2,

    "do_rotation": True,
    "rotation_x": (-15. / 360 * 2. * np.pi, 15. / 360 * 2. * np.pi),
    "rotation_y": (-15. / 360 * 2. * np.pi, 15. / 360 * 2. * np.pi),
    "rotation_z": (-15. / 360 * 2. * np.pi, 15. / 360 * 2. * np.pi),
    "rotation_p_per_axis": 1,
    "p_rot": 0.2,

    "random_crop": False,
    "random_crop_dist_to_border": None,

    "do_gamma": True,
    "gamma_retain_stats": True,
    "gamma_range": (0.7, 1.5),
    "p_gamma": 0.3,

    "do_mirror": True,
    "mirror_axes": (0, 1, 2),

    "dummy_2D": False,
    "mask_was_used_for_normalization": None,
    "border_mode_data": "constant",

    "all_segmentation_labels": None,
    "move_last_seg_chanel_to_data": False,
    "cascade_do_cascade_augmentations": False,

    "do_additive_brightness": False,
    "additive_brightness_p_per_sample": 0.15,
    "additive_brightness_p_per_channel": 0.5,
    "additive_brightness_mu": 0.0,
    "additive_brightness_sigma": 0.1,

    "num_threads": 24 if 'nnUNet_n_proc_DA' not in os.environ else int(os.environ['nnUNet_n_proc_DA']),
    "num_cached_per_thread": 1,
}


def rotation_matrix_3d(angle_x, angle_y, angle_z):
    """batchgenerators create_matrix_rotation_{x,y,z}_3d composed as rotate_coords_3d does: R = Rx Ry Rz, applied to row
    vectors (coords^T R)."""
    cx, sx, cy, sy, cz, sz = np.cos(angle_x), np.sin(angle_x), np.cos(angle_y), np.sin(angle_y), np.cos(angle_z), np.sin(angle_z)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rx @ ry @ rz


def get_patch_size(final_patch_size, rot_x, rot_y, rot_z, scale_range):
    """reference :111-131: the patch the loader must deliver so that every rotation / scaling finds its voxels"""
    def mx(r):
        return max(np.abs(r)) if isinstance(r, (tuple, list)) else r
    rot_x, rot_y, rot_z = (min(90 / 360 * 2. * np.pi, mx(r)) for r in (rot_x, rot_y, rot_z))
    coords = np.array(final_patch_size, dtype=float)
    final_shape = np.copy(coords)
    if len(coords) == 3:
        for ang in ((rot_x, 0, 0), (0, rot_y, 0), (0, 0, rot_z)):
            final_shape = np.max(np.vstack((np.abs(coords @ rotation_matrix_3d(*ang)), final_shape)), 0)
    elif len(coords) == 2:                           # dummy_2D: the in-plane patch of an anisotropic 3D patch (rotate_coords_2d)
        rot = np.array([[np.cos(rot_x), -np.sin(rot_x)], [np.sin(rot_x), np.cos(rot_x)]])
        final_shape = np.max(np.vstack((np.abs(coords @ rot), final_shape)), 0)
    else:
        raise NotImplementedError("patch sizes of %d dimensions" % len(coords))
    final_shape /= min(scale_range)
    return final_shape.astype(int)


default_2D_augmentation_params = deepcopy(default_3D_augmentation_params)
default_2D_augmentation_params["elastic_deform_alpha"] = (0., 200.)
default_2D_augmentation_params["elastic_deform_sigma"] = (9., 13.)
default_2D_augmentation_params["rotation_x"] = (-180. / 360 * 2. * np.pi, 180. / 360 * 2. * np.pi)
default_2D_augmentation_params["rotation_y"] = (-0. / 360 * 2. * np.pi, 0. / 360 * 2. * np.pi)
default_2D_augmentation_params["rotation_z"] = (-0. / 360 * 2. * np.pi, 0. / 360 * 2. * np.pi)
default_2D_augmentation_params["dummy_2D"] = False
default_2D_augmentation_params["mirror_axes"] = (0, 1)
