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

Out of scope (explicit errors): resampling the softmax to another grid (third-party skimage ``resize`` semantics,
segmentation_export.py:84-104) and the NIfTI writer (SimpleITK); ``predict_cases`` therefore takes a ``writer`` callback.
"""
from typing import Callable, Iterable, Optional, Sequence, Tuple

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


def export_segmentation(softmax: torch.Tensor, properties_dict: dict, transpose_backward: Optional[Sequence[int]] = None,
                        region_class_order: Optional[Sequence[int]] = None) -> np.ndarray:
    """uint8 label volume in the ORIGINAL (uncropped) geometry from a device softmax [K, X, Y, Z]
    (reference predict.py:298-301 + segmentation_export.py:73-136 without the resampling branch)."""
    assert softmax.is_cuda and softmax.dtype == torch.float32 and softmax.dim() == 4 and softmax.is_contiguous()
    k = softmax.shape[0]
    tb = [0, 1, 2] if transpose_backward is None else [int(i) for i in transpose_backward]
    dims = [int(softmax.shape[1 + i]) for i in tb]                 # softmax.transpose([0] + [i + 1 for i in tb]).shape[1:]
    strides = [int(softmax.stride(1 + i)) for i in tb]
    after_crop = properties_dict.get('size_after_cropping')
    if after_crop is not None and any(int(a) != int(b) for a, b in zip(dims, after_crop)):
        raise NotImplementedError("the softmax grid %s differs from size_after_cropping %s: resampling to the original spacing "
                                  "(skimage resize, segmentation_export.py:84-104) is outside the MI355X hot path" % (dims, tuple(after_crop)))
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
