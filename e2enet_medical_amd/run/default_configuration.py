"""reference e2enet/run/default_configuration.py:34-80: plans file, output folder, dataset directory, batch-dice switch and stage
of a (network, task) pair.  The trainer class is always ``nnUNetTrainer_simple`` (simple_main.py:145 overrides whatever the lookup
returned)."""
import os
import pickle

from .. import paths
from ..training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple

join = os.path.join


def get_default_configuration(network, task, network_trainer, plans_identifier=paths.default_plans_identifier):
    assert network in ['2d', '3d_lowres', '3d_fullres', '3d_cascade_fullres'], \
        "network can only be one of the following: '2d', '3d_lowres', '3d_fullres', '3d_cascade_fullres'"
    pre = paths.preprocessing_output_dir
    dataset_directory = join(pre, task)
    plans_file = join(pre, task, plans_identifier + ("_plans_2D.pkl" if network == '2d' else "_plans_3D.pkl"))
    with open(plans_file, 'rb') as f:
        plans = pickle.load(f)
    possible_stages = list(plans['plans_per_stage'].keys())
    if (network == '3d_cascade_fullres' or network == "3d_lowres") and len(possible_stages) == 1:
        raise RuntimeError("3d_lowres/3d_cascade_fullres only applies if there is more than one stage. This task does "
                           "not require the cascade. Run 3d_fullres instead")
    stage = 0 if (network == '2d' or network == "3d_lowres") else possible_stages[-1]
    output_folder_name = join(paths.network_training_output_dir, network, task, network_trainer + "__" + plans_identifier)
    print("###############################################")
    print("I am running the following nnUNet: %s" % network)
    print("My trainer class is: ", nnUNetTrainer_simple)
    print("I am using stage %d from these plans" % stage)
    if (network == '2d' or len(possible_stages) > 1) and not network == '3d_lowres':
        batch_dice = True
        print("I am using batch dice + CE loss")
    else:
        batch_dice = False
        print("I am using sample dice + CE loss")
    print("\nI am using data from this folder: ", join(dataset_directory, plans['data_identifier']))
    print("###############################################")
    return plans_file, output_folder_name, dataset_directory, batch_dice, stage, nnUNetTrainer_simple
