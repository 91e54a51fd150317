"""reference e2enet/utilities/task_name_id_conversion.py:20-60"""
import os

import numpy as np

from .. import paths


def _subdirs(folder, prefix):
    if folder is None or not os.path.isdir(folder):
        return []
    return sorted(d for d in os.listdir(folder) if os.path.isdir(os.path.join(folder, d)) and d.startswith(prefix))


def convert_id_to_task_name(task_id: int):
    startswith = "Task%03.0d" % task_id
    cands = _subdirs(paths.preprocessing_output_dir, startswith) + _subdirs(paths.nnUNet_raw_data, startswith) + \
        _subdirs(paths.nnUNet_cropped_data, startswith)
    for m in ['2d', '3d_lowres', '3d_fullres', '3d_cascade_fullres']:
        cands += _subdirs(os.path.join(paths.network_training_output_dir, m), startswith)
    unique = np.unique(cands)
    if len(unique) > 1:
        raise RuntimeError("More than one task name found for task id %d. Please correct that. (I looked in the following "
                           "folders:\n%s\n%s\n%s" % (task_id, paths.nnUNet_raw_data, paths.preprocessing_output_dir,
                                                     paths.nnUNet_cropped_data))
    if len(unique) == 0:
        raise RuntimeError("Could not find a task with the ID %d. Make sure the requested task ID exists and that nnU-Net knows "
                           "where raw and preprocessed data are located. Here are your currently defined folders:\n"
                           "nnUNet_preprocessed=%s\nRESULTS_FOLDER=%s\nnnUNet_raw_data_base=%s"
                           % (task_id, paths.preprocessing_output_dir, paths.network_training_output_dir_base, paths.base))
    return str(unique[0])


def convert_task_name_to_id(task_name: str):
    assert task_name.startswith("Task")
    return int(task_name[4:7])
