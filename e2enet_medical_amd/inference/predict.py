"""Fold ensembling and segmentation export with the softmax volume kept in HBM.

Reference: ``predict_cases`` (e2enet/inference/predict.py:194-359) runs, per case, one sliding-window prediction per
fold, sums the float32 softmax volumes on the HOST (:282-293), divides by the number of folds (:295-296), undoes the
plans' axis transposition (:298-301) and hands the multi-GB array to a worker process that takes the argmax (or the
region thresholds), places the result into the uncropped uint8 volume and writes NIfTI
(``save_segmentation_nifti_from_softmax``, e2enet/inference/segmentation_export.py:27-160).  Here the per-fold
probabilities never leave the device: accumulation, average, transposition, argmax / region thresholds and crop-box
placement are two HIP kernels (``e2e_ensemble_accumulate``, ``e2e_export_argmax_u8``); the host receives the final
uint8 label volume.  Arithmetic is bit-identical to the reference's numpy expressions (float32 adds in fold order,
float32 division by the fold count, first maximum).

Resampling to the grid the case had before preprocessing (segmentation_export.py:84-104: order 1, and order 0 along one
separate low-resolution axis when the spacing is anisotropic beyond ``RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD``) is a third
kernel (``e2e_resample_linear``) with the arithmetic of ``scipy.ndimage.zoom(order=1, mode='nearest', grid_mode=True)`` --
what scikit-image 0.19.3's ``resize(order=1, mode='edge', anti_aliasing=False)`` evaluates.  Parity for that step is against
scipy (oracle/export.py; scikit-image is absent from the image: "parity unpinned").  The NIfTI writer (SimpleITK) is host
tooling; ``predict_cases`` therefore takes a ``writer`` callback.
"""
import os
import pickle
from typing import Callable, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def predict_case_ensemble(trainer, params: Sequence[dict], data: np.ndarray, do_tta: bool = True, step_size: float = 0.5,
                          all_in_gpu: bool = False, mixed_precision: bool = True) -> torch.Tensor:
    """Mean softmax of all folds for one preprocessed case, as a device tensor [K, X, Y, Z]
    (reference predict.py:282-296).  ``params``: the checkpoints of the folds (dicts as torch.load returns them)."""
    net = trainer.network
    keep = net.keep_on_device
    net.keep_on_device = True
    total = None
    try:
        for i, p in enumerate(params):
            trainer.load_checkpoint_ram(p, False)
            _, probs = trainer.predict_preprocessed_data_return_seg_and_softmax(
                data, do_mirroring=do_tta, mirror_axes=trainer.data_aug_params['mirror_axes'], use_sliding_window=True,
                step_size=step_size, use_gaussian=True, all_in_gpu=all_in_gpu, verbose=False, mixed_precision=mixed_precision)
            last = i + 1 == len(params)
            n_div = len(params) if (last and len(params) > 1) else 0
            if total is None:
                total = probs if len(params) == 1 else probs.clone()      # predict_3D reuses nothing, but later folds add in place
            else:
                lib().ensemble_accumulate(total.data_ptr(), probs.data_ptr(), total.numel(), 0, n_div, _stream())
    finally:
        net.keep_on_device = keep
    return total


RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD = 3        # reference e2enet/configuration.py


def get_do_separate_z(spacing, anisotropy_threshold=RESAMPLING_SEPARATE_Z_ANISO_THRESHOLD):
    """reference preprocessing.py:28-30"""
    return (np.max(spacing) / np.min(spacing)) > anisotropy_threshold


def get_lowres_axis(new_spacing):
    """reference preprocessing.py:33-35"""
    return np.where(max(new_spacing) / np.array(new_spacing) == 1)[0]


def resample_plan(properties_dict: dict, force_separate_z: Optional[bool] = None):
    """(do_separate_z, lowres_axis or None): the decision of segmentation_export.py:85-106"""
    if force_separate_z is None:
        if get_do_separate_z(properties_dict.get('original_spacing')):
            do_separate_z, lowres_axis = True, get_lowres_axis(properties_dict.get('original_spacing'))
        elif get_do_separate_z(properties_dict.get('spacing_after_resampling')):
            do_separate_z, lowres_axis = True, get_lowres_axis(properties_dict.get('spacing_after_resampling'))
        else:
            do_separate_z, lowres_axis = False, None
    else:
        do_separate_z = force_separate_z
        lowres_axis = get_lowres_axis(properties_dict.get('original_spacing')) if do_separate_z else None
    if lowres_axis is not None and len(lowres_axis) != 1:
        do_separate_z = False          # spacings like (0.24, 1.25, 1.25): no separate out-of-plane axis (:102-105)
    return do_separate_z, (int(lowres_axis[0]) if do_separate_z else None)


def resample_softmax(softmax: torch.Tensor, new_shape: Sequence[int], transpose_backward: Optional[Sequence[int]] = None,
                     lowres_axis: Optional[int] = None) -> torch.Tensor:
    """Device softmax [K, X, Y, Z] -> contiguous [K] + new_shape in the transposed-backward frame, order 1 (order 0 along
    ``lowres_axis``): resample_data_or_seg(..., is_seg=False, order=1, order_z=0), preprocessing.py:113-202."""
    assert softmax.is_cuda and softmax.dtype == torch.float32 and softmax.dim() == 4
    tb = [0, 1, 2] if transpose_backward is None else [int(i) for i in transpose_backward]
    dims = [int(softmax.shape[1 + i]) for i in tb]
    strides = [int(softmax.stride(1 + i)) for i in tb]
    k = softmax.shape[0]
    new_shape = [int(v) for v in new_shape]
    out = torch.empty([k] + new_shape, dtype=torch.float32, device=softmax.device)
    lib().resample_linear(softmax.data_ptr(), out.data_ptr(), k, int(softmax.stride(0)), dims[0], dims[1], dims[2], strides[0],
                          strides[1], strides[2], new_shape[0], new_shape[1], new_shape[2],
                          -1 if lowres_axis is None else int(lowres_axis), _stream())
    return out


def export_segmentation(softmax: torch.Tensor, properties_dict: dict, transpose_backward: Optional[Sequence[int]] = None,
                        region_class_order: Optional[Sequence[int]] = None, force_separate_z: Optional[bool] = None,
                        order: int = 1, interpolation_order_z: int = 0) -> np.ndarray:
    """uint8 label volume in the ORIGINAL (uncropped) geometry from a device softmax [K, X, Y, Z]
    (reference predict.py:298-301 + segmentation_export.py:73-136)."""
    assert softmax.is_cuda and softmax.dtype == torch.float32 and softmax.dim() == 4 and softmax.is_contiguous()
    k = softmax.shape[0]
    tb = [0, 1, 2] if transpose_backward is None else [int(i) for i in transpose_backward]
    dims = [int(softmax.shape[1 + i]) for i in tb]                 # softmax.transpose([0] + [i + 1 for i in tb]).shape[1:]
    strides = [int(softmax.stride(1 + i)) for i in tb]
    after_crop = properties_dict.get('size_after_cropping')
    if after_crop is not None and any(int(a) != int(b) for a, b in zip(dims, after_crop)):
        # the network ran on a resampled grid: back to the case's own grid before the argmax (segmentation_export.py:84-104)
        if order != 1 or interpolation_order_z != 0:
            raise NotImplementedError("device resampling implements the reference's defaults: order 1 in the plane, order 0 "
                                      "along a separate low-resolution axis (predict.py:171-172)")
        _, lowres = resample_plan(properties_dict, force_separate_z)
        softmax = resample_softmax(softmax, after_crop, tb, lowres)
        dims = [int(v) for v in after_crop]
        strides = [int(softmax.stride(1 + i)) for i in range(3)]
    bbox = properties_dict.get('crop_bbox')
    if bbox is not None:
        out_shape = tuple(int(v) for v in properties_dict.get('original_size_of_raw_data'))
        off = [int(bbox[c][0]) for c in range(3)]
    else:
        out_shape, off = tuple(dims), [0, 0, 0]
    seg = torch.zeros(out_shape, dtype=torch.uint8, device=softmax.device)
    regions = None
    if region_class_order is not None:
        regions = torch.tensor([int(c) for c in region_class_order], dtype=torch.int32, device=softmax.device)
    lib().export_argmax_u8(softmax.data_ptr(), seg.data_ptr(), k, int(softmax.stride(0)), dims[0], dims[1], dims[2], strides[0],
                           strides[1], strides[2], out_shape[0], out_shape[1], out_shape[2], off[0], off[1], off[2],
                           regions.data_ptr() if regions is not None else None, len(regions) if regions is not None else 0,
                           _stream())
    return seg.cpu().numpy()


def predict_cases(trainer, params: Sequence[dict], preprocessed: Iterable[Tuple[str, Tuple[np.ndarray, dict]]],
                  writer: Callable[[np.ndarray, str, dict], None], do_tta: bool = True, step_size: float = 0.5,
                  all_in_gpu: bool = False, mixed_precision: bool = True):
    """The per-case loop of reference predict_cases (predict.py:273-331) on preprocessed cases
    ``(output_filename, (data [C,X,Y,Z], properties_dict))``; ``writer(seg_uint8, output_filename, properties_dict)``
    stores the label volume (the reference's SimpleITK NIfTI writer, segmentation_export.py:144-148, is host tooling)."""
    done = []
    for output_filename, (d, dct) in preprocessed:
        if isinstance(d, str):
            d = np.load(d)
        softmax = predict_case_ensemble(trainer, params, d, do_tta, step_size, all_in_gpu, mixed_precision)
        tb = trainer.plans.get('transpose_backward') if trainer.plans.get('transpose_forward') is not None else None
        seg = export_segmentation(softmax, dct, tb, getattr(trainer, 'regions_class_order', None))
        writer(seg, output_filename, dct)
        done.append(output_filename)
    return done


def nifti_writer():
    """writer(seg_uint8, path, properties) of predict_cases / predict_from_folder: the reference's SimpleITK export
    (segmentation_export.py:144-148) when SimpleITK is importable, else ``<case>.npy`` beside the requested ``.nii.gz`` path."""
    try:
        import SimpleITK as sitk
    except ImportError:
        sitk = None

    def write(seg, path, props):
        if sitk is None:
            np.save(path[:-7] + ".npy" if path.endswith(".nii.gz") else path + ".npy", seg)
            return
        img = sitk.GetImageFromArray(seg.astype(np.uint8))
        img.SetSpacing(props['itk_spacing'])
        img.SetOrigin(props['itk_origin'])
        img.SetDirection(props['itk_direction'])
        sitk.WriteImage(img, path)
    return write


def check_input_folder_and_return_caseIDs(input_folder: str, expected_num_modalities: int):
    """Case identifiers of an input folder.  Two layouts are served:
      * PREPROCESSED cases (what the reference's GenericPreprocessor writes): ``<case>.npz`` (or ``<case>.npy``) holding
        [modalities (+ seg), X, Y, Z] plus ``<case>.pkl`` with the properties dict -- returned as (ids, "preprocessed");
      * the reference's raw layout ``<case>_XXXX.nii.gz`` (predict.py:631-672) -- returned as (ids, "nifti"); predicting from it
        needs the reference's preprocessing package (crop, resample, normalise: SimpleITK / scikit-image), outside this engine."""
    files = sorted(os.listdir(input_folder))
    nii = [f for f in files if f.endswith(".nii.gz")]
    if nii:
        ids = sorted({f[:-12] for f in nii})
        missing = [c + "_%04.0d.nii.gz" % n for c in ids for n in range(expected_num_modalities)
                   if not os.path.isfile(os.path.join(input_folder, c + "_%04.0d.nii.gz" % n))]
        if missing:
            print("Some files are missing:")
            print(missing)
            raise RuntimeError("missing files in input_folder")
        return ids, "nifti"
    ids = sorted({f[:-4] for f in files if (f.endswith(".npz") or (f.endswith(".npy") and not f.endswith("_segs.npy")))
                  and os.path.isfile(os.path.join(input_folder, f[:-4] + ".pkl"))})
    return ids, "preprocessed"


def predict_from_folder(model: str, input_folder: str, output_folder: str, folds: Union[Tuple[int], List[int], None],
                        save_npz: bool, num_threads_preprocessing: int, num_threads_nifti_save: int,
                        lowres_segmentations: Union[str, None], part_id: int, num_parts: int, tta: bool,
                        mixed_precision: bool = True, overwrite_existing: bool = True, mode: str = 'normal',
                        overwrite_all_in_gpu: bool = None, step_size: float = 0.5,
                        checkpoint_name: str = "model_final_checkpoint", segmentation_export_kwargs: dict = None,
                        disable_postprocessing: bool = False, writer=None):
    """reference predict.py:675-764 with the same arguments: the cases ``[part_id::num_parts]`` of ``input_folder`` through
    ``load_model_and_checkpoint_files`` (model_restore.py:108-154) -> fold ensemble -> export, written to ``output_folder``.
    The per-case work is ``predict_cases`` above (softmax resident in HBM).  ``num_threads_*`` are accepted and unused: there are
    no preprocessing / export worker processes here."""
    import shutil
    from ..training.model_restore import load_model_and_checkpoint_files
    if mode != "normal":
        raise NotImplementedError("mode=%r: the engine implements the reference's 'normal' mode (predict.py:729-737)" % (mode,))
    if lowres_segmentations is not None:
        raise NotImplementedError("cascade inputs (lowres_segmentations) are outside the shiftConvPP hot path")
    os.makedirs(output_folder, exist_ok=True)
    assert os.path.isfile(os.path.join(model, "plans.pkl")), "Folder with saved model weights must contain a plans.pkl file"
    shutil.copy(os.path.join(model, 'plans.pkl'), output_folder)
    with open(os.path.join(model, "plans.pkl"), 'rb') as f:
        expected_num_modalities = pickle.load(f)['num_modalities']
    case_ids, layout = check_input_folder_and_return_caseIDs(input_folder, expected_num_modalities)
    if layout == "nifti":
        raise NotImplementedError(
            "%s holds raw NIfTI cases (<case>_XXXX.nii.gz): cropping / resampling / normalisation is the reference's preprocessing "
            "package (e2enet/preprocessing, SimpleITK + scikit-image; SURVEY.md section 2 row 11, out of scope).  Preprocess the "
            "folder with it and point -i at the resulting <case>.npz + <case>.pkl files" % input_folder)
    case_ids = case_ids[part_id::num_parts]
    output_files = [os.path.join(output_folder, c + ".nii.gz") for c in case_ids]
    if not overwrite_existing:
        keep = [i for i, o in enumerate(output_files) if not (os.path.isfile(o) or os.path.isfile(o[:-7] + ".npy"))]
        case_ids, output_files = [case_ids[i] for i in keep], [output_files[i] for i in keep]
    print("number of cases that still need to be predicted:", len(case_ids))
    if not case_ids:
        return []
    trainer, params = load_model_and_checkpoint_files(model, folds, mixed_precision=mixed_precision, checkpoint_name=checkpoint_name)
    all_in_gpu = False if overwrite_all_in_gpu is None else overwrite_all_in_gpu

    def cases():
        for c, out in zip(case_ids, output_files):
            npy, npz = os.path.join(input_folder, c + ".npy"), os.path.join(input_folder, c + ".npz")
            d = np.load(npy) if os.path.isfile(npy) else np.load(npz)['data']
            with open(os.path.join(input_folder, c + ".pkl"), 'rb') as f:
                props = pickle.load(f)
            d = np.asarray(d)[:expected_num_modalities]          # (preprocessed training cases carry their labels as a last channel)
            yield out, (d, props)
    return predict_cases(trainer, params, cases(), writer if writer is not None else nifti_writer(), do_tta=tta,
                         step_size=step_size, all_in_gpu=all_in_gpu, mixed_precision=mixed_precision)
