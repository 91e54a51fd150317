"""Trainer + checkpoint restore for inference (reference e2enet/training/model_restore.py:44-154)."""
import os
import pickle

import torch

join, isdir = os.path.join, os.path.isdir


def restore_model(pkl_file, checkpoint=None, train=False, fp16=None):
    """reference :44-99: instantiate the trainer named in ``<checkpoint>.pkl`` with its saved constructor arguments.  The only
    trainer of this package is ``nnUNetTrainer_simple`` (simple_main.py:145 forces it in the reference as well); a pickle naming
    another class raises."""
    from .network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
    with open(pkl_file, 'rb') as f:
        info = pickle.load(f)
    if info['name'] != nnUNetTrainer_simple.__name__:
        raise RuntimeError("Could not find the model trainer specified in checkpoint: %r (this package provides %s).\n"
                           "Debug info: \ncheckpoint file: %s" % (info['name'], nnUNetTrainer_simple.__name__, checkpoint))
    trainer = nnUNetTrainer_simple(*info['init'])
    if info.get('e2e_base_num_features') is not None:
        # (the reference hard-codes width 48, nnUNetTrainer_simple.py:296; a trainer of this package that was given another width
        #  records it next to its constructor arguments)
        trainer.base_num_features_override = info['e2e_base_num_features']
    if fp16 is not None:
        trainer.fp16 = False            # (accepted and ignored: the engine computes in fp32)
    trainer.process_plans(info['plans'])
    if checkpoint is not None:
        trainer.load_checkpoint(checkpoint, train)
    return trainer


def load_model_and_checkpoint_files(folder, folds=None, mixed_precision=None, checkpoint_name="model_best"):
    """reference :108-154: restore the trainer from fold 0's pickle, take ``Tconv`` from the checkpoint name (:145-148), initialize
    for inference and load every fold's parameters to host memory."""
    if isinstance(folds, str):
        folds = [join(folder, "all")]
        assert isdir(folds[0]), "no output folder for fold %s found" % folds
    elif isinstance(folds, (list, tuple)):
        if len(folds) == 1 and folds[0] == "all":
            folds = [join(folder, "all")]
        else:
            folds = [join(folder, "fold_%d" % i) for i in folds]
        assert all([isdir(i) for i in folds]), "list of folds specified but not all output folders are present"
    elif isinstance(folds, int):
        folds = [join(folder, "fold_%d" % folds)]
        assert all([isdir(i) for i in folds]), "output folder missing for fold %d" % folds
    elif folds is None:
        print("folds is None so we will automatically look for output folders (not using 'all'!)")
        folds = sorted(join(folder, d) for d in os.listdir(folder) if d.startswith("fold") and isdir(join(folder, d)))
        print("found the following folds: ", folds)
    else:
        raise ValueError("Unknown value for folds. Type: %s. Expected: list of int, int, str or None" % str(type(folds)))
    trainer = restore_model(join(folds[0], "%s.model.pkl" % checkpoint_name), fp16=mixed_precision)
    trainer.output_folder = folder
    trainer.output_folder_base = folder
    trainer.update_fold(0)
    if 'shiftConvPP' in checkpoint_name:
        trainer.Tconv = checkpoint_name.split('_model')[0]
    else:
        trainer.Tconv = 'ori'           # (another architecture: initialize_network raises NotImplementedError naming it)
    trainer.initialize(False)
    all_best_model_files = [join(i, "%s.model" % checkpoint_name) for i in folds]
    print("using the following model files: ", all_best_model_files)
    all_params = [torch.load(i, map_location=torch.device('cpu'), weights_only=False) for i in all_best_model_files]
    return trainer, all_params
