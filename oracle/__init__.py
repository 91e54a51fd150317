"""CPU oracle for the E2ENet shiftConvPP + DSFF hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (torch-CPU / numpy,
fp32) of the reference algorithm.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
``e2enet_medical_amd`` never does and fails loudly when its HIP library is
missing.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_*``
against golden vectors produced by importing the reference itself
(``tools/make_golden.py`` -> ``tests/golden/*.npz``) and against the reference's
own known-answer vectors for the sliding-window tile placement
(reference tests/test_steps_for_sliding_window_prediction.py:96-163).

Third-party arithmetic boundary (not under /root/reference): conv / instance
norm / pooling / transposed conv / softmax / CE / SGD / CosineAnnealingLR are
PyTorch operators (reference pins torch==1.12.1, requirements.txt:56),
``gaussian_filter`` is scipy (==1.9.1, :49), ``pad_nd_image`` is batchgenerators
(==0.24, :1).  The oracle calls the same operators from the torch/scipy in this
image (operator semantics identical, last-ulp drift possible).
"""
from .shift import shift_amounts, depth_shift                      # noqa: F401
from .network import (NetSpec, make_spec, param_shapes, init_params,  # noqa: F401
                      forward, conv_block, masked_names, Branches)
from .dsff import (CosineDeathRate, uniform_kernel_masks, kernel_l1,   # noqa: F401
                   kernel_death, kernel_growth, kernel_grad_growth, DsffState)
from .sliding_window import (compute_steps, gaussian_map, pad_to_patch,  # noqa: F401
                             mirror_predict, predict_tiled)
from .loss import dc_ce_loss, deep_supervision_loss, ds_weights, hard_dice  # noqa: F401
from .optim import clip_and_sgd_step, poly_lr                          # noqa: F401
from .export import ensemble_softmax, export_segmentation                  # noqa: F401
from .ds_targets import downsample_seg_for_ds                              # noqa: F401  (parity unpinned: see its header)
