"""``python -m e2enet_medical_amd.simple_main`` -- the reference's training entry point (simple_main.py:33-216) on the MI355X
engine: the same argv (incl. the sparse flags of ``add_sparse_args``), the same sequence

    get_default_configuration -> nnUNetTrainer_simple(...) -> initialize() -> CosineDecay / Masking / add_module
    -> [load_latest_checkpoint | pretrained weights] -> run_training(mask)      |  --validation_only: load checkpoint -> validate

Differences, all stated: ``--fp32`` / mixed precision is accepted and ignored (fp32 results); ``-c`` restores the DSFF masks and the
optimizer state from the checkpoint (the reference re-draws the masks before loading); ``--base_num_features``
is an extension (the reference hard-codes 48, nnUNetTrainer_simple.py:296); ``--synthetic_data`` trains on seeded noise when no
preprocessed data exists (benchmarks / smoke tests); ``--validation_only`` really validates (the reference's call is commented out,
:200-208); ``--find_lr`` and the 3d_lowres cascade hand-over are outside the engine."""
import argparse

import torch

from . import paths
from .run.default_configuration import get_default_configuration
from .training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
from .training.network_training.sparselearning.core_channel import Masking, CosineDecay, add_sparse_args
from .utilities.task_name_id_conversion import convert_id_to_task_name


def str2bool(s):
    return True if s.lower() == 'true' else False


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--network", type=str, default='3d_fullres')
    parser.add_argument("--network_trainer", type=str, default='nnUNetTrainerV2')
    parser.add_argument("--task", help="can be task name or task id", type=str, default='1')
    parser.add_argument("--fold", help='0, 1, ..., 5 or \'all\'', type=str, default='0')
    parser.add_argument("--validation_only", required=False, type=str2bool, default=False,
                        help="use this if you want to only run the validation")
    parser.add_argument("-c", "--continue_training", help="use this if you want to continue a training", action="store_true")
    parser.add_argument("-p", help="plans identifier. Only change this if you created a custom experiment planner",
                        default=paths.default_plans_identifier, required=False)
    parser.add_argument("--use_compressed_data", default=False, action="store_true", required=False)
    parser.add_argument("--deterministic", required=False, default=False, action="store_true")
    parser.add_argument("--npz", required=False, default=True, action="store_true")
    parser.add_argument("--find_lr", required=False, default=False, action="store_true", help="not used here, just for fun")
    parser.add_argument("--valbest", required=False, default=False, type=str2bool)
    parser.add_argument("--fp32", required=False, default=False, action="store_true",
                        help="disable mixed precision training and run old school fp32 (the engine always does)")
    parser.add_argument("--val_folder", required=False, default="validation_raw")
    parser.add_argument("--disable_saving", required=False, action='store_true')
    parser.add_argument("--disable_postprocessing_on_folds", required=False, action='store_true')
    parser.add_argument('--val_disable_overwrite', action='store_false', default=True,
                        help='Validation does not overwrite existing segmentations')
    parser.add_argument('--disable_next_stage_pred', action='store_true', default=False, help='do not predict next stage')
    parser.add_argument('--pretrained_weights', type=str, required=False, default=None)
    parser.add_argument('--Tconv', type=str, required=False, default='ori', help='ori;shiftConvPP')
    parser.add_argument('--max_num_epochs', type=int, required=False, default=5)
    parser.add_argument('--num_batches_per_epoch', type=int, required=False, default=5)
    # ---- extensions of this package (absent from the reference's parser) ----
    parser.add_argument('--base_num_features', type=int, required=False, default=None,
                        help='network width; default: the reference\'s hard-coded 48')
    parser.add_argument('--synthetic_data', action='store_true', default=False,
                        help='train on seeded Gaussian noise / random labels when no preprocessed data exists')
    add_sparse_args(parser)
    return parser


def load_pretrained_weights(network, fname, verbose=False):
    """reference e2enet/run/load_pretrained_weights.py: copy every checkpoint tensor whose name and shape match the network
    (segmentation heads may differ between tasks and are skipped)."""
    saved = torch.load(fname, map_location=torch.device('cpu'), weights_only=False)['state_dict']
    own = network.state_dict()
    new = {}
    for k, v in saved.items():
        key = k[7:] if k.startswith('module.') else k
        if key in own and tuple(own[key].shape) == tuple(v.shape):
            new[key] = v
    if not new:
        raise RuntimeError("Pretrained weights are not compatible with the current network architecture")
    own.update(new)
    print("################### Loading pretrained weights from file ", fname, '###################')
    network.load_state_dict(own)


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    print(args)
    task, fold, network = args.task, args.fold, args.network
    if not task.startswith("Task"):
        task = convert_id_to_task_name(int(task))
    if fold != 'all':
        fold = int(fold)
    plans_file, output_folder_name, dataset_directory, batch_dice, stage, trainer_class = \
        get_default_configuration(network, task, args.network_trainer, args.p)
    trainer_class = nnUNetTrainer_simple                                                 # simple_main.py:145
    trainer = trainer_class(plans_file, fold, output_folder=output_folder_name, dataset_directory=dataset_directory,
                            batch_dice=batch_dice, stage=stage, unpack_data=not args.use_compressed_data,
                            deterministic=args.deterministic, fp16=not args.fp32, Tconv=args.Tconv,
                            max_num_epochs=args.max_num_epochs, num_batches_per_epoch=args.num_batches_per_epoch, args=args)
    trainer.base_num_features_override = args.base_num_features
    trainer.synthetic_data = bool(args.synthetic_data)
    if args.disable_saving:
        trainer.save_final_checkpoint = False
        trainer.save_best_checkpoint = False
        trainer.save_intermediate_checkpoints = True
        trainer.save_latest_only = True
    model, optimizer = trainer.initialize(not args.validation_only)
    print("Total parameters count", sum(p.numel() for p in model.parameters() if p.requires_grad))
    mask = None
    if args.sparse:
        decay = CosineDecay(args.death_rate, args.max_num_epochs * args.num_batches_per_epoch)
        mask = Masking(optimizer, death_rate=args.death_rate, death_mode=args.death, death_rate_decay=decay,
                       growth_mode=args.growth, redistribution_mode=args.redistribution, args=args)
        mask.add_module(model, sparse_init=args.sparse_init, density=args.density)
    if args.find_lr:
        raise NotImplementedError("--find_lr (reference nnUNetTrainer_simple.find_lr, 'just for fun') is outside the engine")
    if not args.validation_only:
        if args.continue_training:
            # (the checkpoint written by run_training carries 'dsff_state': masks, schedule position and RNG continue where they were)
            try:
                trainer.load_latest_checkpoint(mask=mask)
            except KeyError:
                trainer.load_latest_checkpoint()          # a checkpoint of the reference: no DSFF state, fresh masks like the reference
        elif args.pretrained_weights is not None:
            load_pretrained_weights(trainer.network, args.pretrained_weights)
        trainer.run_training(mask)
    else:
        if args.valbest:
            trainer.load_best_checkpoint(train=False)
        else:
            trainer.load_final_checkpoint(train=False)
    trainer.network.eval()
    if args.validation_only:
        trainer.validate(save_softmax=args.npz, validation_folder_name=args.val_folder,
                         run_postprocessing_on_folds=not args.disable_postprocessing_on_folds, overwrite=args.val_disable_overwrite)
    print(f'finish training {args.Tconv} !!!')
    if network == '3d_lowres' and not args.disable_next_stage_pred:
        raise NotImplementedError("predict_next_stage (3d_lowres cascade) is outside the shiftConvPP hot path")
    return trainer


if __name__ == "__main__":
    main()
