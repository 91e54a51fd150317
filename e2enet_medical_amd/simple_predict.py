"""``python -m e2enet_medical_amd.simple_predict`` -- the reference's inference entry point (simple_predict.py:25-228) on the MI355X
engine: the same argv; the checkpoint name is prefixed with ``--Tconv`` (:152) and ``predict_from_folder`` runs the cases
``[part_id::num_parts]`` (one process per GPU; within a process ``SegmentationNetwork.shard_tiles`` can split the tiles of a case
over a process group instead).  The input folder holds PREPROCESSED cases (``<case>.npz|.npy`` + ``<case>.pkl``): raw NIfTI needs
the reference's preprocessing package (``inference.predict.predict_from_folder`` says so)."""
import argparse
import os

from . import paths
from .inference.predict import predict_from_folder
from .utilities.task_name_id_conversion import convert_id_to_task_name

join, isdir = os.path.join, os.path.isdir


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-i", '--input_folder', required=False,
                        default='../nnUNet/nnUNet_raw_data_base/nnUNet_raw_data/Task216_AMOS2022_task1/imagesTs/')
    parser.add_argument('-o', "--output_folder", required=False, help="folder for saving predictions",
                        default='../nnUnet/results_imageTs/output_task216_d07_best_new')
    parser.add_argument('-t', '--task_name', help='task name or task ID, required.', default='17', required=False)
    parser.add_argument('-tr', '--trainer_class_name', required=False, default=paths.default_trainer)
    parser.add_argument('-ctr', '--cascade_trainer_class_name', required=False, default=paths.default_cascade_trainer)
    parser.add_argument('-m', '--model', default="3d_fullres", required=False)
    parser.add_argument('-p', '--plans_identifier', default=paths.default_plans_identifier, required=False)
    parser.add_argument('-f', '--folds', nargs='+', default='None')
    parser.add_argument('-z', '--save_npz', required=False, action='store_true')
    parser.add_argument('-l', '--lowres_segmentations', required=False, default='None')
    parser.add_argument("--part_id", type=int, required=False, default=0)
    parser.add_argument("--num_parts", type=int, required=False, default=1)
    parser.add_argument("--num_threads_preprocessing", required=False, default=6, type=int)
    parser.add_argument("--num_threads_nifti_save", required=False, default=2, type=int)
    parser.add_argument("--disable_tta", required=False, default=False, action="store_true")
    parser.add_argument("--overwrite_existing", required=False, default=False, action="store_true")
    parser.add_argument("--mode", type=str, default="normal", required=False, help="Hands off!")
    parser.add_argument("--all_in_gpu", type=str, default="False", required=False)
    parser.add_argument("--step_size", type=float, default=0.5, required=False, help="don't touch")
    parser.add_argument('-chk', help='checkpoint name, default: model_final_checkpoint', required=False,
                        default='model_final_checkpoint')
    parser.add_argument('--disable_mixed_precision', default=False, action='store_true', required=False)
    parser.add_argument('--Tconv', type=str, required=False, default='shiftConvPP', help='ori;shiftConvPP')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    folds, lowres_segmentations, all_in_gpu, model = args.folds, args.lowres_segmentations, args.all_in_gpu, args.model
    task_name = args.task_name
    args.chk = f"{args.Tconv}_{args.chk}"                       # simple_predict.py:152
    print(args.chk)
    if not task_name.startswith("Task"):
        task_name = convert_id_to_task_name(int(task_name))
    if lowres_segmentations == "None":
        lowres_segmentations = None
    if isinstance(folds, list):
        if not (folds[0] == 'all' and len(folds) == 1):
            folds = [int(i) for i in folds]
    elif folds == "None":
        folds = None
    else:
        raise ValueError("Unexpected value for argument folds")
    assert all_in_gpu in ['None', 'False', 'True']
    all_in_gpu = {"None": None, "True": True, "False": False}[all_in_gpu]
    if model == "3d_cascade_fullres":
        raise NotImplementedError("the 3d_cascade_fullres model is outside the shiftConvPP hot path")
    model_folder_name = join(paths.network_training_output_dir, model, task_name, args.trainer_class_name + "__" + args.plans_identifier)
    print("using model stored in ", model_folder_name)
    assert isdir(model_folder_name), "model output folder not found. Expected: %s" % model_folder_name
    return predict_from_folder(model_folder_name, args.input_folder, args.output_folder, folds, args.save_npz,
                               args.num_threads_preprocessing, args.num_threads_nifti_save, lowres_segmentations, args.part_id,
                               args.num_parts, not args.disable_tta, overwrite_existing=args.overwrite_existing, mode=args.mode,
                               overwrite_all_in_gpu=all_in_gpu, mixed_precision=not args.disable_mixed_precision,
                               step_size=args.step_size, checkpoint_name=args.chk)


if __name__ == "__main__":
    main()
