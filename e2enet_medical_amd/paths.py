"""Folder conventions of the reference (e2enet/paths.py:18-66).

The reference hard-codes three relative folders (:33-35: ``nnUNet_raw_data_base``, ``nnUNet_preprocessed``,
``results_e2enet_ch48_216_133_shift5``) and keeps the nnU-Net environment variables as commented-out lines (:29-31).  Here the
environment variables win when they are set and the reference's folder names are the defaults, so a checkout of the reference with
its data in place and an nnU-Net style environment both work unchanged.  Nothing is created at import time (the reference calls
``maybe_mkdir_p`` while importing)."""
import os

my_output_identifier = "nnUNet"
default_plans_identifier = "nnUNetPlansv2.1"
default_data_identifier = 'nnUNetData_plans_v2.1'
default_trainer = "nnUNetTrainerV2"
default_cascade_trainer = "nnUNetTrainerV2CascadeFullRes"


def _folders():
    base = os.environ.get('nnUNet_raw_data_base', 'nnUNet_raw_data_base')
    pre = os.environ.get('nnUNet_preprocessed', 'nnUNet_preprocessed')
    res = os.environ.get('RESULTS_FOLDER', 'results_e2enet_ch48_216_133_shift5')
    return base, pre, res


def __getattr__(name):          # resolved at access time: tests and launchers set the environment after importing the package
    base, pre, res = _folders()
    if name == "base":
        return base
    if name == "preprocessing_output_dir":
        return pre
    if name == "network_training_output_dir_base":
        return res
    if name == "network_training_output_dir":
        return os.path.join(res, my_output_identifier)
    if name == "nnUNet_raw_data":
        return os.path.join(base, "nnUNet_raw_data")
    if name == "nnUNet_cropped_data":
        return os.path.join(base, "nnUNet_cropped_data")
    raise AttributeError(name)
