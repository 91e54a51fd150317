"""``nnUNetTrainer_simple`` on the MI355X engine.

Keeps the operator surface of reference e2enet/training/network_training/nnUNetTrainer_simple.py that ``simple_main.py``,
``simple_predict.py`` and ``model_restore.load_model_and_checkpoint_files`` touch:

  constructor (:59-176)                 initialize (:178-253)            initialize_network (:255-363, shiftConvPP :292-301)
  initialize_optimizer_and_scheduler (:365-369)                          run_iteration (:529-583)
  run_online_evaluation / finish_online_evaluation (:371-423)            update_fold (:653-680)
  maybe_update_lr / maybe_save_checkpoint (:758-785)                     update_eval_criterion_MA (:787-810)
  manage_patience (:812-860)            on_epoch_end (:863-877)          update_train_loss_MA (:879-884)
  run_training (:929-1027)              process_plans (:1036-1103)       save_checkpoint (:1140-1176)
  load_best / latest / final_checkpoint (:1178-1202)                     load_checkpoint(_ram) (:1204-1255)
  predict_preprocessed_data_return_seg_and_softmax (:491-527)           validate (:1309-1479)

The arithmetic of an iteration -- forward, deep-supervision Dice+CE, backward, clip_grad_norm_(12), SGD-Nesterov, DSFF
mask, and for validation batches the loss value and the hard tp/fp/fn counts -- is one chain of HIP kernel launches
(``engine.Engine`` + ``fused_optim.FusedClipSGD``), fp32 throughout.  When ``torch.distributed`` is initialised the
iteration is data parallel (one process per GPU, RCCL): bucketed gradient all-reduce overlapped with the backward pass,
global batch dice, broadcast DSFF masks (``parallel``; the reference's collective inventory is nnUNetTrainerV2_DDP.py:198,
:263-268).

Out of scope: the batchgenerators data pipeline and preprocessing: those entry points raise
NotImplementedError naming the subsystem.  ``validate`` (:1309-1479) runs on the engine; NIfTI reading / writing is a callback
(SimpleITK when it is importable).  Data arrives through any iterator of
``{'data': [B,C,...], 'target': [list of [B,1,...] per scale]}`` dicts (the format of the reference's augmenter output,
:538-540); ``SyntheticGenerator`` provides seeded synthetic batches.
"""
import os
import pickle
from collections import OrderedDict
from datetime import datetime
from time import time
from typing import Tuple

import numpy as np
import torch
from torch import nn

from ...network_architecture.initialization import InitWeights_He
from ...network_architecture.neural_network import SegmentationNetwork
from ...network_architecture.unetpp_d import Generic_UNetPlusPlus
from ...utilities.nd_softmax import softmax_helper
from ..fused_optim import FusedClipSGD
from ..learning_rate.poly_lr import poly_lr
from ..loss_functions.deep_supervision import MultipleOutputLoss2
from ..loss_functions.dice_loss import DC_and_CE_loss

join, isfile = os.path.join, os.path.isfile


class SyntheticGenerator:
    """Seeded synthetic batches shaped like the reference augmenter's output (data N(0,1), nearest-downsampled
    integer targets per deep-supervision scale)."""

    def __init__(self, batch_size, channels, patch_size, num_classes, ds_scales, seed=0, device="cuda"):
        self.shape = (batch_size, channels) + tuple(int(p) for p in patch_size)
        self.k, self.scales, self.dev = num_classes, ds_scales, device
        self.gen = torch.Generator().manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        data = torch.randn(self.shape, generator=self.gen)
        full = torch.randint(0, self.k, (self.shape[0], 1) + self.shape[2:], generator=self.gen).float()
        targets = []
        for sc in self.scales:
            step = [max(1, int(round(1 / s))) for s in sc]
            targets.append(full[:, :, ::step[0], ::step[1], ::step[2]].contiguous())
        return {'data': data, 'target': targets}

    next = __next__                      # the reference calls tr_gen.next() once before training (:946-947)


def _out_of_scope(what, where):
    raise NotImplementedError("%s needs the reference's %s, which is outside the MI355X hot path (SURVEY.md section 2, "
                              "OUT OF SCOPE rows); run it with the reference package on the outputs of this trainer"
                              % (what, where))


class nnUNetTrainer_simple(object):
    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False, Tconv=None, max_num_epochs=200,
                 num_batches_per_epoch=100, args=None):
        self.init_args = (plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data,
                          deterministic, fp16)
        self.deep_supervision_scales = self.ds_loss_weights = None
        self.pin_memory = True
        self.args = args
        # the engine computes in fp32 (1e-4 logit parity bar); the reference's AMP switch is accepted and ignored
        self.fp16 = False
        self.amp_grad_scaler = None
        self.unpack_data, self.stage = unpack_data, stage
        self.experiment_name = self.__class__.__name__
        self.plans_file, self.output_folder, self.dataset_directory = plans_file, output_folder, dataset_directory
        self.output_folder_base = self.output_folder
        self.output_folder_pretrained = self.output_folder
        self.fold = fold
        self.plans = None
        self.Tconv = Tconv if Tconv is not None else 'shiftConvPP'
        self.gt_niftis_folder = join(dataset_directory, "gt_segmentations") \
            if dataset_directory is not None and os.path.isdir(dataset_directory) else None
        self.folder_with_preprocessed_data = None
        self.dl_tr = self.dl_val = None
        self.num_input_channels = self.num_classes = self.net_pool_per_axis = self.patch_size = self.batch_size = \
            self.threeD = self.base_num_features = self.intensity_properties = self.normalization_schemes = \
            self.net_num_pool_op_kernel_sizes = self.net_conv_kernel_sizes = None
        self.basic_generator_patch_size = self.transpose_forward = self.transpose_backward = None
        self.data_aug_params = {'mirror_axes': (0, 1, 2), 'do_mirror': True}
        self.batch_dice = batch_dice
        self.loss = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []
        self.classes = self.do_dummy_2D_aug = self.use_mask_for_norm = self.only_keep_largest_connected_component = \
            self.min_region_size_per_class = self.min_size_per_class = None
        self.inference_pad_border_mode = "constant"
        self.inference_pad_kwargs = {'constant_values': 0}
        self.update_fold(fold)
        self.pad_all_sides = None
        self.lr_scheduler_eps, self.lr_scheduler_patience = 1e-3, 30
        self.initial_lr, self.weight_decay = 1e-2, 3e-5
        self.oversample_foreground_percent = 0.33
        self.conv_per_stage = None
        self.regions_class_order = None
        self.network = self.optimizer = self.lr_scheduler = None
        self.tr_gen = self.val_gen = None
        self.was_initialized = False
        self.dataset = self.dataset_tr = self.dataset_val = None
        self.patience = 50
        self.val_eval_criterion_alpha = 0.9
        self.train_loss_MA_alpha = 0.93
        self.train_loss_MA_eps = 5e-4
        self.max_num_epochs, self.num_batches_per_epoch = max_num_epochs, num_batches_per_epoch
        self.num_val_batches_per_epoch = 50
        self.also_val_in_tr_mode = False
        self.lr_threshold = 1e-6
        self.val_eval_criterion_MA = self.train_loss_MA = None
        self.best_val_eval_criterion_MA = self.best_MA_tr_loss_for_patience = self.best_epoch_based_on_MA_tr_loss = None
        self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics = [], [], [], []
        self.epoch = 0
        self.log_file = None
        self.deterministic = deterministic
        self.use_progress_bar = bool(int(os.environ.get('nnunet_use_progress_bar', '0')))
        self.save_every = 50
        self.save_latest_only = True
        self.save_intermediate_checkpoints = True
        self.save_best_checkpoint = True
        self.save_final_checkpoint = True
        # ---- this engine ----
        self.base_num_features_override = None      # reference hard-codes 48 for shiftConvPP (:296)
        self.synthetic_data = False                 # True: initialize() may install SyntheticGenerator (never silently)
        self.prefetch_batches = True                # fetch + upload batch i+1 under the GPU work of batch i (see run_iteration)
        self._prefetched, self._copy_stream = {}, None
        self._prefetch_error = {}                   # id(generator) -> (generator, exception raised by its one-ahead fetch)
        self.process_group = None                   # torch.distributed group of the data-parallel replicas (None = world)
        self._fused = None
        self._dp = {}                               # id(engine) -> OverlappedGradAllReduce
        self._mask = None                           # the Masking passed to run_training (checkpoints carry its state)

    # ------------------------------------------------------------------------------------------ folds / plans
    def update_fold(self, fold):
        """reference :653-680: swap between folds for inference (ensembles of cross-validation models)."""
        if fold is not None:
            of = self.output_folder
            if isinstance(fold, str):
                assert fold == "all", "if self.fold is a string then it must be 'all'"
                if of is not None:
                    if of.endswith("%s" % str(self.fold)):
                        of = self.output_folder_base
                    of = join(of, "%s" % str(fold))
            elif of is not None:
                if of.endswith("fold_%s" % str(self.fold)):
                    of = self.output_folder_base
                of = join(of, "fold_%s" % str(fold))
            self.output_folder = of
            self.fold = fold
            self.output_folder_pretrained = self.output_folder      # reference :679-680 (its Tconv branch is commented out)

    def load_plans_file(self):
        if isinstance(self.plans_file, dict):
            self.plans = self.plans_file
        else:
            with open(self.plans_file, 'rb') as f:
                self.plans = pickle.load(f)

    def process_plans(self, plans):
        """reference :1036-1103 (the keys the E2ENet path consumes)."""
        if self.stage is None:
            assert len(list(plans['plans_per_stage'].keys())) == 1, \
                "If self.stage is None then there can be only one stage in the plans file"
            self.stage = list(plans['plans_per_stage'].keys())[0]
        self.plans = plans
        sp = plans['plans_per_stage'][self.stage]
        self.batch_size = sp['batch_size']
        self.patch_size = np.array(sp['patch_size']).astype(int)
        self.net_num_pool_op_kernel_sizes = sp['pool_op_kernel_sizes']
        self.net_conv_kernel_sizes = sp.get('conv_kernel_sizes')
        self.do_dummy_2D_aug = sp.get('do_dummy_2D_data_aug', False)
        self.base_num_features = plans['base_num_features']
        self.num_input_channels = plans['num_modalities']
        self.num_classes = plans['num_classes'] + 1
        self.classes = plans.get('all_classes')
        self.intensity_properties = plans.get('dataset_properties', {}).get('intensityproperties') \
            if isinstance(plans.get('dataset_properties'), dict) else None
        self.normalization_schemes = plans.get('normalization_schemes')
        self.use_mask_for_norm = plans.get('use_mask_for_norm')
        self.transpose_forward = plans.get('transpose_forward', [0, 1, 2])
        self.transpose_backward = plans.get('transpose_backward', [0, 1, 2])
        self.conv_per_stage = plans.get('conv_per_stage', 2)
        self.threeD = len(self.patch_size) == 3
        if not self.threeD:
            raise RuntimeError("the MI355X engine implements 3d_fullres plans only")

    def setup_DA_params(self):
        """reference :682-733: deep-supervision target scales from the cumulative pooling, rotations of +-30 degrees, scale range
        (0.7, 1.4), no elastic deformation; `do_dummy_2D_data_aug` plans (anisotropic patches) rotate in-plane only, by up to
        180 degrees, and the loader's patch keeps the network's depth."""
        from ..data_augmentation.default_data_augmentation import (default_3D_augmentation_params, default_2D_augmentation_params,
                                                                   get_patch_size)
        self.deep_supervision_scales = [[1, 1, 1]] + list(list(i) for i in 1 / np.cumprod(
            np.vstack(self.net_num_pool_op_kernel_sizes), axis=0))[:-1]
        p = dict(default_3D_augmentation_params)             # (a copy: the reference mutates the module-level table)
        ang = (-30. / 360 * 2. * np.pi, 30. / 360 * 2. * np.pi)
        p['rotation_x'] = p['rotation_y'] = p['rotation_z'] = ang
        if self.do_dummy_2D_aug:
            p["dummy_2D"] = True
            p["elastic_deform_alpha"] = default_2D_augmentation_params["elastic_deform_alpha"]
            p["elastic_deform_sigma"] = default_2D_augmentation_params["elastic_deform_sigma"]
            p["rotation_x"] = default_2D_augmentation_params["rotation_x"]
        p["mask_was_used_for_normalization"] = self.use_mask_for_norm
        if self.do_dummy_2D_aug:
            bg = get_patch_size(self.patch_size[1:], p['rotation_x'], p['rotation_y'], p['rotation_z'], p['scale_range'])
            self.basic_generator_patch_size = np.array([self.patch_size[0]] + list(bg))
        else:
            self.basic_generator_patch_size = get_patch_size(self.patch_size, p['rotation_x'], p['rotation_y'], p['rotation_z'],
                                                             p['scale_range'])
        p["scale_range"] = (0.7, 1.4)
        p["do_elastic"] = False
        p['selected_seg_channels'] = [0]
        p['patch_size_for_spatialtransform'] = self.patch_size
        p["num_cached_per_thread"] = 2
        self.data_aug_params = p

    # ------------------------------------------------------------------------------------------ initialize
    def initialize(self, training=True, force_load_plans=False):
        if not self.was_initialized:
            if self.output_folder:
                os.makedirs(self.output_folder, exist_ok=True)
            if force_load_plans or (self.plans is None):
                self.load_plans_file()
            self.process_plans(self.plans)
            self.setup_DA_params()
            net_numpool = len(self.net_num_pool_op_kernel_sizes)
            weights = np.array([1 / (2 ** i) for i in range(net_numpool)])          # reference :200-213
            mask = np.array([True] + [True if i < net_numpool - 1 else False for i in range(1, net_numpool)])
            weights[~mask] = 0
            self.ds_loss_weights = weights / weights.sum()
            self.loss = MultipleOutputLoss2(self.loss, self.ds_loss_weights)
            self.initialize_network()
            self.initialize_optimizer_and_scheduler()
            if training and self.tr_gen is None:
                scales = self.deep_supervision_scales[:self._num_ds_outputs()]
                folder = None
                if self.dataset_directory is not None and 'data_identifier' in self.plans:
                    folder = join(self.dataset_directory, self.plans['data_identifier'] + "_stage%d" % self.stage)     # reference :216-217
                if folder is not None and os.path.isdir(folder):
                    # reference :218-239: DataLoader3D x 2 over the preprocessed cases + the moreDA chain (here: on the device)
                    from ..data_augmentation.data_augmentation_moreDA import get_moreDA_augmentation
                    self.folder_with_preprocessed_data = folder
                    self.dl_tr, self.dl_val = self.get_basic_generators()
                    self.tr_gen, self.val_gen = get_moreDA_augmentation(
                        self.dl_tr, self.dl_val, self.data_aug_params['patch_size_for_spatialtransform'], self.data_aug_params,
                        deep_supervision_scales=scales, pin_memory=self.pin_memory, use_nondetMultiThreadedAugmenter=False)
                    self.print_to_log_file("TRAINING KEYS LEN: %s" % (len(self.dataset_tr.keys())), also_print_to_console=False)
                    self.print_to_log_file("VALIDATION KEYS LEN: %s" % (len(self.dataset_val.keys())), also_print_to_console=False)
                elif self.synthetic_data:
                    dev = "cuda" if torch.cuda.is_available() else "cpu"
                    self.print_to_log_file("WARNING: synthetic_data=True: training on seeded Gaussian noise and random labels "
                                           "(benchmarks and smoke tests only)")
                    self.tr_gen = SyntheticGenerator(self.batch_size, self.num_input_channels, self.patch_size,
                                                     self.num_classes, scales, seed=0, device=dev)
                    self.val_gen = SyntheticGenerator(self.batch_size, self.num_input_channels, self.patch_size,
                                                      self.num_classes, scales, seed=1, device=dev)
                else:
                    raise FileNotFoundError(
                        "no preprocessed data under %r (dataset_directory / plans['data_identifier'] + '_stage%s', reference "
                        "nnUNetTrainer_simple.py:216-217): preprocess the task with the reference package, hand the trainer its own "
                        "tr_gen / val_gen before initialize(), or set trainer.synthetic_data = True to train on synthetic noise on "
                        "purpose" % (folder, self.stage))
            assert isinstance(self.network, (SegmentationNetwork, nn.DataParallel))
        self.was_initialized = True
        return self.network, self.optimizer

    def initialize_network(self):
        """reference :292-301 (Tconv == 'shiftConvPP') and its kernel-shape ablations 'shiftConvPP_313' / 'shiftConvPP_331'
        (:303-323, same constructor arguments), 'shiftConvPP_noshift' (:337-346: the (1,3,3) network without the shift) and
        'shiftConvPP_nodff' (:326-335: the plain U-Net wiring of the same blocks); the remaining Tconv values are other
        architectures outside this engine."""
        extra = {}
        if self.Tconv == 'shiftConvPP':
            net_cls = Generic_UNetPlusPlus
        elif self.Tconv == 'shiftConvPP_313':
            from ...network_architecture.unetpp_d_313 import Generic_UNetPlusPlus as net_cls
        elif self.Tconv == 'shiftConvPP_331':
            from ...network_architecture.unetpp_d_331 import Generic_UNetPlusPlus as net_cls
        elif self.Tconv == 'shiftConvPP_noshift':
            net_cls, extra = Generic_UNetPlusPlus, {"shift_size": 1}
        elif self.Tconv == 'shiftConvPP_nodff':
            from ...network_architecture.unetpp_d_nodff import Generic_UNetPlusPlus as net_cls
        else:
            raise NotImplementedError("the MI355X engine implements Tconv='shiftConvPP' and its ablations 'shiftConvPP_313', "
                                      "'shiftConvPP_331', 'shiftConvPP_noshift', 'shiftConvPP_nodff' (got %r)" % (self.Tconv,))
        base = 48 if self.base_num_features_override is None else self.base_num_features_override
        self.network = net_cls(self.patch_size, self.num_input_channels, base, self.num_classes,
                                            len(self.net_num_pool_op_kernel_sizes), self.conv_per_stage, 2, nn.Conv3d,
                                            nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                                            {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                            {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x,
                                            InitWeights_He(1e-2), self.net_num_pool_op_kernel_sizes,
                                            self.net_conv_kernel_sizes, False, True, True, **extra)
        if torch.cuda.is_available():
            self.network.cuda()
        self.network.inference_apply_nonlin = softmax_helper

    def initialize_optimizer_and_scheduler(self):
        assert self.network is not None, "self.initialize_network must be called first"
        self.optimizer = torch.optim.SGD(self.network.parameters(), self.initial_lr, weight_decay=self.weight_decay,
                                         momentum=0.99, nesterov=True)
        self.lr_scheduler = None
        self._fused = None

    # ------------------------------------------------------------------------------------------ data parallel
    def _data_parallel(self):
        """(active, group): data-parallel replicas exist when torch.distributed is initialised with more than one rank
        (E2E_FORCE_DIST=1 runs the same code path on a single rank: self-test of the collectives)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False, None
        force = os.environ.get("E2E_FORCE_DIST") == "1"
        return dist.get_world_size(self.process_group) > 1 or force, self.process_group

    def _num_ds_outputs(self):
        """deep-supervision outputs the network emits = targets the loss consumes (one per segmentation head; the
        reference's plans give num_pool scales, of which the heads cover the first num_pool - 1, unetpp_d.py:394-401)"""
        heads = getattr(self.network, "seg_outputs", None)
        n = len(heads) if heads is not None else len(self.deep_supervision_scales)
        assert self.ds_loss_weights is None or all(w == 0 for w in self.ds_loss_weights[n:]), \
            "non-zero deep-supervision loss weight beyond the network's %d heads" % n
        return min(n, len(self.deep_supervision_scales))

    def _rank(self):
        """(rank, world, group) of the data-parallel job; (0, 1, None) for a single process"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0, 1, None
        return dist.get_rank(self.process_group), dist.get_world_size(self.process_group), self.process_group

    def _rank_mean(self, value):
        """mean of a host scalar over the data-parallel ranks: epoch losses feed the moving averages and the patience
        decision, which every rank must take identically (reference nnUNetTrainerV2_DDP keeps them on rank 0's view)"""
        rank, world, group = self._rank()
        if world == 1:
            return float(value)
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=next(self.network.parameters()).device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return float(t.item()) / world

    def _dp_for(self, eng, group):
        from ... import parallel
        dp = self._dp.get(id(eng))
        if dp is None:
            eng.prepare_backward()
            dp = parallel.OverlappedGradAllReduce(eng, group=group, force=os.environ.get("E2E_FORCE_DIST") == "1")
            eng.batch_dice_hook = parallel.batch_dice_allreduce(group)
            for old in self._dp.values():                        # a plan evicted from the network's cache: drop its hooks
                old_eng = getattr(old, "engine", None)
                if old_eng is not None and old_eng is not eng:
                    old_eng.grad_bucket_hook = None
                    old_eng.batch_dice_hook = None
            self._dp = {id(eng): dp}
        return dp

    # ------------------------------------------------------------------------------------------ one iteration
    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, mask=None):
        """reference :529-583.  Returns the loss as a numpy scalar (one device->host sync, like the reference)."""
        dev = next(self.network.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("nnUNetTrainer_simple (MI355X) needs the network on a GPU: there is no CPU fallback")
        # the batch: uploaded under the previous iteration when it was fetched one call ahead (prefetch_batches), else now
        err = self._prefetch_error.pop(id(data_generator), None)
        if err is not None:
            raise err[1]        # the generator failed while the previous step (already applied, loss returned) fetched one ahead
        slot = self._prefetched.pop(id(data_generator), None) if self.prefetch_batches else None
        if slot is None:
            data, target = self._upload(next(data_generator), dev)
        else:
            _gen, data, target, ev = slot                       # (_gen: the strong reference kept its id() from being reused)
            torch.cuda.current_stream().wait_event(ev)
        if len(target) == 1 and self.deep_supervision_scales is not None and len(self.deep_supervision_scales) > 1:
            # a generator that yields the full-resolution labels only: the deep-supervision scales are gathered on the
            # device (the reference's DownsampleSegForDSTransform2 step of the CPU augmentation pipeline, N3)
            from ..data_augmentation.downsampling import downsample_seg_for_ds_transform2
            target = downsample_seg_for_ds_transform2(target[0], self.deep_supervision_scales[:self._num_ds_outputs()], order=0)
        if getattr(self.network, "conv_variant", "133") != "133":     # kernel-shape ablations run on axis-permuted tensors
            data = self.network.to_engine_layout(data)
            target = [self.network.to_engine_layout(t) for t in target]
        eng = self.network.engine(data)
        dp_on, group = self._data_parallel()
        dp = self._dp_for(eng, group) if (dp_on and do_backprop) else None
        if dp_on and eng.batch_dice_hook is None:
            from ... import parallel
            eng.batch_dice_hook = parallel.batch_dice_allreduce(group)
        eng.forward(data, deep_supervision=True)
        if do_backprop:
            loss = eng.loss_backward(target, self.ds_loss_weights, batch_dice=self.batch_dice)
            if dp is not None:
                dp.finish()                                     # gradients averaged over the replicas (clip norm sees these)
            if self._fused is None:
                self._fused = FusedClipSGD(self.optimizer, list(self.network.named_parameters()), max_norm=12.0)
            masks = mask.masks if mask is not None else None
            self._fused.step(eng.grads, masks)                  # clip + SGD + (weight, momentum) *= mask
            if mask is not None:
                if dp_on and getattr(mask, "process_group", None) is None:
                    mask.process_group = group
                if getattr(mask, "growth_mode", "random") == "gradient":
                    # the reference's kernel_grad_growth reads weight.grad after clip_grad_norm_ (core_channel.py:833-835)
                    mask.set_gradients(eng.grads, self._fused._sq, self._fused.max_norm)
                mask.step(masks_already_applied=True)
        else:
            loss = eng.loss_value(target, self.ds_loss_weights, batch_dice=self.batch_dice)
        if run_online_evaluation:
            self.run_online_evaluation([h.out.data for h in eng.heads], target, _engine=eng)
        if self.prefetch_batches:
            # the step's kernels are queued; fetch the NEXT batch of this generator and upload it on a copy stream while they
            # run (the reference uploads synchronously at the top of run_iteration: 86 MB = 2 ms of the 52 ms iteration
            # at the benchmarked shape).  The generator is therefore consumed one batch ahead.
            try:
                nxt = next(data_generator)
            except StopIteration:
                nxt = None
            except Exception as e:                               # this step is applied: report on the next call for this generator
                self._prefetch_error[id(data_generator)] = (data_generator, e)
                nxt = None
            if nxt is not None:
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(device=dev)
                main = torch.cuda.current_stream()
                with torch.cuda.stream(self._copy_stream):
                    d2, t2 = self._upload(nxt, dev)
                    ev = torch.cuda.Event()
                    ev.record(self._copy_stream)
                for t in [d2] + list(t2):
                    t.record_stream(main)
                if self._has_pinned(nxt):
                    ev.synchronize()                             # a generator may refill its pinned buffers in place on the next fetch
                while len(self._prefetched) >= 4:               # generators that were dropped by the caller: forget their batch
                    self._prefetched.pop(next(iter(self._prefetched)))
                self._prefetched[id(data_generator)] = (data_generator, d2, t2, ev)
        return loss.detach().cpu().numpy().reshape(())

    def drain_prefetched(self):
        """Forget the batches fetched one call ahead (run_iteration consumes a generator one batch ahead of the step it
        returns; call this when a generator is replaced and its pending batch must not be used)."""
        self._prefetched.clear()
        self._prefetch_error.clear()

    @staticmethod
    def _has_pinned(data_dict):
        vals = [data_dict['data']] + (list(data_dict['target']) if isinstance(data_dict['target'], (list, tuple)) else [data_dict['target']])
        return any(isinstance(v, torch.Tensor) and v.is_pinned() for v in vals)

    @staticmethod
    def _upload(data_dict, dev):
        """host batch -> float32 device tensors on the current stream (reference maybe_to_torch + to_cuda, :533-542)"""
        data, target = data_dict['data'], data_dict['target']
        data = torch.as_tensor(data).float().to(dev, non_blocking=True).contiguous()
        if not isinstance(target, (list, tuple)):
            target = [target]
        target = [torch.as_tensor(t).float().to(dev, non_blocking=True).contiguous() for t in target]
        return data, target

    def run_online_evaluation(self, output, target, _engine=None):
        """reference :371-405: hard tp/fp/fn per foreground class of the full-resolution prediction, summed over the
        batch (HIP kernel e2e_online_eval_counts; under data parallelism summed over the ranks as
        nnUNetTrainerV2_DDP.py:303-305 does)."""
        if _engine is not None:
            counts = _engine.online_eval_counts(target[0])
        else:
            # the reference's public signature (foreign logits [B, K, ...] + labels): counted straight from those tensors,
            # no activation plan is built for them
            from ..._lib import lib
            logits, tgt = output[0], target[0]
            if getattr(self.network, "conv_variant", "133") != "133":
                logits, tgt = self.network.to_engine_layout(logits), self.network.to_engine_layout(tgt)
            logits, tgt = logits.float().contiguous(), tgt.float().contiguous()
            if not logits.is_cuda:
                raise RuntimeError("run_online_evaluation (MI355X) needs GPU tensors: there is no CPU fallback")
            b, k = logits.shape[:2]
            spatial = logits[0, 0].numel()
            assert tgt.numel() == b * spatial, "target must hold one label per voxel of the logits"
            counts = torch.zeros((k, 3), dtype=torch.int64, device=logits.device)
            lib().online_eval_counts(logits.data_ptr(), tgt.data_ptr(), counts.data_ptr(), b, k, spatial,
                                     torch.cuda.current_stream().cuda_stream)
        dp_on, group = self._data_parallel()
        if dp_on:
            import torch.distributed as dist
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        c = counts.cpu().numpy().astype(np.float32)[1:]          # foreground classes
        tp_hard, fp_hard, fn_hard = c[:, 0], c[:, 1], c[:, 2]
        self.online_eval_foreground_dc.append(list((2 * tp_hard) / (2 * tp_hard + fp_hard + fn_hard + 1e-8)))
        self.online_eval_tp.append(list(tp_hard))
        self.online_eval_fp.append(list(fp_hard))
        self.online_eval_fn.append(list(fn_hard))

    def finish_online_evaluation(self):
        """reference :407-423"""
        if len(self.online_eval_tp) == 0:
            return
        self.online_eval_tp = np.sum(self.online_eval_tp, 0)
        self.online_eval_fp = np.sum(self.online_eval_fp, 0)
        self.online_eval_fn = np.sum(self.online_eval_fn, 0)
        with np.errstate(divide='ignore', invalid='ignore'):
            global_dc_per_class = [i for i in [2 * i / (2 * i + j + k) for i, j, k in
                                               zip(self.online_eval_tp, self.online_eval_fp, self.online_eval_fn)]
                                   if not np.isnan(i)]
        self.all_val_eval_metrics.append(np.mean(global_dc_per_class))
        self.print_to_log_file("Average global foreground Dice:", [np.round(i, 4) for i in global_dc_per_class])
        self.print_to_log_file("(interpret this as an estimate for the Dice of the different classes. This is not exact.)")
        self.online_eval_foreground_dc, self.online_eval_tp, self.online_eval_fp, self.online_eval_fn = [], [], [], []

    # ------------------------------------------------------------------------------------------ epoch bookkeeping
    def maybe_update_lr(self, epoch=None):
        """reference :758-773 (called in on_epoch_end before the epoch counter is incremented, hence the +1)."""
        ep = self.epoch + 1 if epoch is None else epoch
        self.optimizer.param_groups[0]['lr'] = poly_lr(ep, self.max_num_epochs, self.initial_lr, 0.9)
        self.print_to_log_file("lr:", np.round(self.optimizer.param_groups[0]['lr'], decimals=6))

    def maybe_save_checkpoint(self):
        """reference :775-785"""
        if self.output_folder and self.save_intermediate_checkpoints and (self.epoch % self.save_every == (self.save_every - 1)):
            self.print_to_log_file("saving scheduled checkpoint file...")
            if not self.save_latest_only:
                self.save_checkpoint(join(self.output_folder, "model_ep_%03.0d.model" % (self.epoch + 1)), mask=self._mask)
            self.save_checkpoint(join(self.output_folder, "%s_model_latest.model" % self.Tconv), mask=self._mask)
            self.print_to_log_file("done")

    def update_eval_criterion_MA(self):
        """reference :787-810"""
        if self.val_eval_criterion_MA is None:
            self.val_eval_criterion_MA = - self.all_val_losses[-1] if len(self.all_val_eval_metrics) == 0 \
                else self.all_val_eval_metrics[-1]
        elif len(self.all_val_eval_metrics) == 0:
            self.val_eval_criterion_MA = self.val_eval_criterion_alpha * self.val_eval_criterion_MA - (
                1 - self.val_eval_criterion_alpha) * self.all_val_losses[-1]
        else:
            self.val_eval_criterion_MA = self.val_eval_criterion_alpha * self.val_eval_criterion_MA + (
                1 - self.val_eval_criterion_alpha) * self.all_val_eval_metrics[-1]

    def update_train_loss_MA(self):
        """reference :879-884"""
        if self.train_loss_MA is None:
            self.train_loss_MA = self.all_tr_losses[-1]
        else:
            self.train_loss_MA = self.train_loss_MA_alpha * self.train_loss_MA + (1 - self.train_loss_MA_alpha) * \
                self.all_tr_losses[-1]

    def manage_patience(self):
        """reference :812-860"""
        continue_training = True
        if self.patience is not None:
            if self.best_MA_tr_loss_for_patience is None:
                self.best_MA_tr_loss_for_patience = self.train_loss_MA
            if self.best_epoch_based_on_MA_tr_loss is None:
                self.best_epoch_based_on_MA_tr_loss = self.epoch
            if self.best_val_eval_criterion_MA is None:
                self.best_val_eval_criterion_MA = self.val_eval_criterion_MA
            self.print_to_log_file("current best_val_eval_criterion_MA is %.4f" % self.best_val_eval_criterion_MA)
            self.print_to_log_file("current val_eval_criterion_MA is %.4f" % self.val_eval_criterion_MA)
            if self.val_eval_criterion_MA > self.best_val_eval_criterion_MA:
                self.best_val_eval_criterion_MA = self.val_eval_criterion_MA
                self.print_to_log_file("saving best epoch checkpoint...")
                if self.save_best_checkpoint and self.output_folder:
                    self.save_checkpoint(join(self.output_folder, "%s_model_best.model" % self.Tconv), mask=self._mask)
            if self.train_loss_MA + self.train_loss_MA_eps < self.best_MA_tr_loss_for_patience:
                self.best_MA_tr_loss_for_patience = self.train_loss_MA
                self.best_epoch_based_on_MA_tr_loss = self.epoch
                self.print_to_log_file("New best epoch (train loss MA): %03.4f" % self.best_MA_tr_loss_for_patience)
            if self.epoch - self.best_epoch_based_on_MA_tr_loss > self.patience:
                if self.optimizer.param_groups[0]['lr'] > self.lr_threshold:
                    self.best_epoch_based_on_MA_tr_loss = self.epoch - self.patience // 2
                else:
                    self.print_to_log_file("My patience ended")
                    continue_training = False
        return continue_training

    def on_epoch_end(self):
        """reference :863-877"""
        self.finish_online_evaluation()
        self.maybe_update_lr()
        self.maybe_save_checkpoint()
        self.update_eval_criterion_MA()
        return self.manage_patience()

    def run_training(self, mask=None):
        """reference :929-1027: epoch loop, validation batches with online evaluation, poly LR, moving averages, best /
        latest / final checkpoints.  Checkpoints written from here carry the DSFF state of ``mask``."""
        self._mask = mask
        self.maybe_update_lr(self.epoch)
        ds = self.network.do_ds
        self.network.do_ds = True
        if not self.was_initialized:
            self.initialize(True)
        if self.output_folder:
            os.makedirs(self.output_folder, exist_ok=True)
            if self.output_folder_base and self._rank()[0] == 0:
                # what predict_from_folder reads the modality count from (inference/predict.py:704-707).  The reference writes it in
                # save_debug_information (:906), which simple_main.py leaves commented out (:186)
                os.makedirs(self.output_folder_base, exist_ok=True)
                with open(join(self.output_folder_base, "plans.pkl"), 'wb') as f:
                    pickle.dump(self.plans, f)
        while self.epoch < self.max_num_epochs:
            self.print_to_log_file("\nepoch: ", self.epoch)
            epoch_start_time = time()
            self.network.train()
            train_losses_epoch = [self.run_iteration(self.tr_gen, True, mask=mask) for _ in range(self.num_batches_per_epoch)]
            self.all_tr_losses.append(self._rank_mean(np.mean(train_losses_epoch)))
            self.print_to_log_file("train loss : %.4f" % self.all_tr_losses[-1])
            with torch.no_grad():
                self.network.eval()
                val_losses = [self.run_iteration(self.val_gen, False, True) for _ in range(self.num_val_batches_per_epoch)]
                self.all_val_losses.append(self._rank_mean(np.mean(val_losses)))
                self.print_to_log_file("validation loss: %.4f" % self.all_val_losses[-1])
            self.update_train_loss_MA()
            continue_training = self.on_epoch_end()
            if not continue_training:
                break
            self.epoch += 1
            self.print_to_log_file("This epoch took %f s\n" % (time() - epoch_start_time))
        self.epoch -= 1          # reference :1018: otherwise the final checkpoint stores an epoch beyond the loss history
        if self.output_folder:
            if self.save_final_checkpoint:
                self.save_checkpoint(join(self.output_folder, "%s_model_final_checkpoint.model" % self.Tconv), mask=mask)
            if self._rank()[0] == 0:
                for f in ("%s_model_latest.model" % self.Tconv, "%s_model_latest.model.pkl" % self.Tconv):
                    if isfile(join(self.output_folder, f)):          # identical with final (reference :1022-1025)
                        os.remove(join(self.output_folder, f))
        self.network.do_ds = ds

    def print_to_log_file(self, *args, also_print_to_console=True, add_timestamp=True):
        """reference :1106-1138 (timestamped text log next to the checkpoints)."""
        if self._rank()[0] != 0:                   # data parallel: one log, written by rank 0 (nnUNetTrainerV2_DDP.py: local_rank == 0)
            return
        if self.output_folder and self.log_file is None:
            ts = datetime.now()
            self.log_file = join(self.output_folder, "training_log_%d_%d_%d_%02.0d_%02.0d_%02.0d.txt" %
                                 (ts.year, ts.month, ts.day, ts.hour, ts.minute, ts.second))
        if self.log_file is not None:
            try:
                with open(self.log_file, 'a+') as f:
                    if add_timestamp:
                        f.write("%s: " % datetime.now())
                    f.write(" ".join(str(a) for a in args) + "\n")
            except IOError:
                pass
        if also_print_to_console:
            print(*args)

    # ------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, fname, save_optimizer=True, mask=None, collective=True):
        """reference :1140-1176 (same dict keys and the side-car .pkl).  With ``mask`` (a Masking) the checkpoint also
        carries 'dsff_state' (packed kernel maps, death-rate schedule position, growth RNG state); reference loaders
        ignore the extra key.

        Under data parallelism this call is COLLECTIVE by default: every rank must make it; rank 0 writes, the others wait,
        and the outcome is broadcast so that a failed write (disk full, permissions) raises on every rank rather than leaving
        the others in a barrier.  ``collective=False`` is the escape for callers that guard the call with ``if rank == 0``:
        the calling rank writes and no other rank is involved."""
        mask = mask if mask is not None else self._mask
        rank, world, group = self._rank()
        if world > 1 and collective:
            # replicas hold identical weights, optimizer state and masks: rank 0 writes (every rank writing the same path at
            # once can leave a torn .model / .pkl), then one broadcast carries success / failure to everybody
            import torch.distributed as dist
            err = None
            if rank == 0:
                try:
                    self._write_checkpoint(fname, save_optimizer, mask)
                except Exception as e:          # noqa: BLE001 -- reported on every rank below, re-raised on rank 0
                    err = e
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
            # (src is a GLOBAL rank: group rank 0 of a sub-group is not global rank 0 in general)
            dist.broadcast(ok, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if err is not None:
                raise err
            if int(ok.item()) != 1:
                raise RuntimeError("save_checkpoint(%s): rank 0 failed to write the checkpoint" % fname)
            return
        self._write_checkpoint(fname, save_optimizer, mask)

    def _write_checkpoint(self, fname, save_optimizer, mask):
        state_dict = OrderedDict((k, v.cpu()) for k, v in self.network.state_dict().items())
        save_this = {'epoch': self.epoch + 1, 'state_dict': state_dict,
                     'optimizer_state_dict': self.optimizer.state_dict() if save_optimizer else None,
                     'lr_scheduler_state_dict': None,
                     'plot_stuff': (self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode,
                                    self.all_val_eval_metrics),
                     'best_stuff': (self.best_epoch_based_on_MA_tr_loss, self.best_MA_tr_loss_for_patience,
                                    self.best_val_eval_criterion_MA)}
        if mask is not None:
            save_this['dsff_state'] = mask.state_dict()
        torch.save(save_this, fname)
        info = OrderedDict(init=self.init_args, name=self.__class__.__name__, plans=self.plans)
        info['class'] = str(self.__class__)
        if getattr(self, 'base_num_features_override', None) is not None:
            info['e2e_base_num_features'] = self.base_num_features_override      # (extra key: reference loaders ignore it)
        with open(fname + ".pkl", 'wb') as f:
            pickle.dump(info, f)

    def load_best_checkpoint(self, train=True, mask=None):
        """reference :1178-1186"""
        if self.fold is None:
            raise RuntimeError("Cannot load best checkpoint if self.fold is None")
        f = join(self.output_folder_pretrained, "%s_model_best.model" % self.Tconv)
        if isfile(f):
            return self.load_checkpoint(f, train=train, mask=mask)
        self.print_to_log_file("WARNING! model_best.model does not exist! Cannot load best checkpoint. Falling "
                               "back to load_latest_checkpoint")
        return self.load_latest_checkpoint(train, mask=mask)

    def load_latest_checkpoint(self, train=True, mask=None):
        """reference :1188-1195.  The reference looks for un-prefixed names only and therefore cannot find what its own
        run_training writes (``{Tconv}_model_latest.model``); both spellings are tried here, the
        prefixed one first."""
        folder = self.output_folder_pretrained
        for stem in ("model_final_checkpoint.model", "model_latest.model"):
            for name in ("%s_%s" % (self.Tconv, stem), stem):
                if isfile(join(folder, name)):
                    return self.load_checkpoint(join(folder, name), train=train, mask=mask)
        if isfile(join(folder, "%s_model_best.model" % self.Tconv)) or isfile(join(folder, "model_best.model")):
            return self.load_best_checkpoint(train, mask=mask)
        raise RuntimeError("No checkpoint found")

    def load_final_checkpoint(self, train=False, mask=None):
        """reference :1197-1202"""
        filename = join(self.output_folder_pretrained, "%s_model_final_checkpoint.model" % self.Tconv)
        if not isfile(filename):
            raise RuntimeError("Final checkpoint not found. Expected: %s. Please finish the training first." % filename)
        return self.load_checkpoint(filename, train=train, mask=mask)

    def load_checkpoint(self, fname, train=True, mask=None):
        self.print_to_log_file("loading checkpoint", fname, "train=", train)
        if not self.was_initialized:
            self.initialize(train)
        self.load_checkpoint_ram(torch.load(fname, map_location=torch.device('cpu'), weights_only=False), train, mask)

    def load_checkpoint_ram(self, checkpoint, train=True, mask=None):
        """reference :1211-1255; ``mask``: a Masking already attached to this trainer's network/optimizer
        (add_module done) that is restored from checkpoint['dsff_state'].  Unlike the reference, the optimizer state is
        restored too when training resumes (the reference restarts the momentum from zero)."""
        if not self.was_initialized:
            self.initialize(train)
        keys = list(self.network.state_dict().keys())
        new_state_dict = OrderedDict()
        for k, value in checkpoint['state_dict'].items():
            key = k[7:] if (k not in keys and k.startswith('module.')) else k
            new_state_dict[key] = value
        self.network.load_state_dict(new_state_dict)
        self.epoch = checkpoint['epoch']
        if train and checkpoint.get('optimizer_state_dict') is not None:
            self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
            self._fused = None
        self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics = \
            checkpoint['plot_stuff']
        if 'best_stuff' in checkpoint:
            self.best_epoch_based_on_MA_tr_loss, self.best_MA_tr_loss_for_patience, self.best_val_eval_criterion_MA = \
                checkpoint['best_stuff']
        if self.epoch != len(self.all_tr_losses):               # reference :1245-1253 (old off-by-one of the final save)
            self.print_to_log_file("WARNING in loading checkpoint: self.epoch != len(self.all_tr_losses). "
                                   "self.epoch is now set to len(self.all_tr_losses)")
            self.epoch = len(self.all_tr_losses)
            self.all_tr_losses = self.all_tr_losses[:self.epoch]
            self.all_val_losses = self.all_val_losses[:self.epoch]
            self.all_val_losses_tr_mode = self.all_val_losses_tr_mode[:self.epoch]
            self.all_val_eval_metrics = self.all_val_eval_metrics[:self.epoch]
        if mask is not None:
            if 'dsff_state' not in checkpoint:
                raise KeyError("checkpoint carries no 'dsff_state' (written by save_checkpoint(..., mask=mask))")
            mask.load_state_dict(checkpoint['dsff_state'])
        # DSFF checkpoints carry their masks implicitly (pruned kernels are exact zeros): skip them at inference
        if not train:
            self.network.enable_auto_sparsity(True)

    # ------------------------------------------------------------------------------------------ inference
    def predict_preprocessed_data_return_seg_and_softmax(self, data: np.ndarray, do_mirroring: bool = True,
                                                         mirror_axes: Tuple[int] = None, use_sliding_window: bool = True,
                                                         step_size: float = 0.5, use_gaussian: bool = True,
                                                         pad_border_mode: str = 'constant', pad_kwargs: dict = None,
                                                         all_in_gpu: bool = False, verbose: bool = True,
                                                         mixed_precision=True) -> Tuple[np.ndarray, np.ndarray]:
        """reference :491-527"""
        ds = self.network.do_ds
        self.network.do_ds = False
        if pad_border_mode == 'constant' and pad_kwargs is None:
            pad_kwargs = {'constant_values': 0}
        if do_mirroring and mirror_axes is None:
            mirror_axes = self.data_aug_params['mirror_axes']
        if do_mirroring:
            assert self.data_aug_params["do_mirror"], "Cannot do mirroring as test time augmentation when training " \
                                                      "was done without mirroring"
        assert isinstance(self.network, (SegmentationNetwork, nn.DataParallel))
        current_mode = self.network.training
        self.network.eval()
        ret = self.network.predict_3D(data, do_mirroring=do_mirroring, mirror_axes=mirror_axes,
                                      use_sliding_window=use_sliding_window, step_size=step_size,
                                      patch_size=tuple(int(p) for p in self.patch_size),
                                      regions_class_order=self.regions_class_order, use_gaussian=use_gaussian,
                                      pad_border_mode=pad_border_mode, pad_kwargs=pad_kwargs, all_in_gpu=all_in_gpu,
                                      verbose=verbose, mixed_precision=mixed_precision)
        self.network.train(current_mode)
        self.network.do_ds = ds
        return ret

    # ------------------------------------------------------------------------------------------ validation
    def validate(self, do_mirroring: bool = True, use_sliding_window: bool = True, step_size: float = 0.5,
                 save_softmax: bool = True, use_gaussian: bool = True, overwrite: bool = True,
                 validation_folder_name: str = 'validation_raw', debug: bool = False, all_in_gpu: bool = False,
                 segmentation_export_kwargs: dict = None, run_postprocessing_on_folds: bool = True, writer=None, gt_reader=None):
        """reference :1309-1479: every case of the validation split through the sliding-window prediction, exported to the case's
        original geometry, scored against the ground truth, ``summary.json`` written in the reference's structure.

        On this engine the softmax volume never leaves the device between prediction and export (``inference.predict.
        export_segmentation``: transpose_backward, resampling, argmax, crop-box placement as HIP kernels); the host receives the
        uint8 label volume.  NIfTI I/O is host tooling (SimpleITK, absent from this image):
          ``writer(seg_uint8, path_nii_gz, properties)``   default: SimpleITK when importable, else ``<case>.npy`` next to it;
          ``gt_reader(path_nii_gz) -> label array``         default: SimpleITK when importable, else ``<case>.npy`` in the ground-truth
                                                           folder, else the segmentation channel of the preprocessed case (then the
                                                           comparison happens on the network's grid and summary.json says so).
        The connected-component post-processing search (``determine_postprocessing``, e2enet/postprocessing) is outside the hot
        path and is skipped with a log line.  Returns the score dict ``aggregate_scores`` builds (the reference returns None)."""
        import json
        import shutil
        from ...inference.predict import export_segmentation
        from ...evaluation.evaluator import aggregate_scores
        current_mode = self.network.training
        self.network.eval()
        assert self.was_initialized, "must initialize, ideally with checkpoint (or train first)"
        if self.dataset_val is None:
            if self.folder_with_preprocessed_data is None:
                if self.dataset_directory is None or 'data_identifier' not in self.plans:
                    raise FileNotFoundError("validate() needs the preprocessed cases of the task (dataset_directory / plans["
                                            "'data_identifier'] + '_stage%s', reference nnUNetTrainer_simple.py:216-217); this trainer "
                                            "has none (synthetic batches?)" % (self.stage,))
                self.folder_with_preprocessed_data = join(self.dataset_directory, self.plans['data_identifier'] + "_stage%d" % self.stage)
            self.load_dataset()
            self.do_split()
        if segmentation_export_kwargs is None:
            if 'segmentation_export_params' in self.plans.keys():
                force_separate_z = self.plans['segmentation_export_params']['force_separate_z']
                interpolation_order = self.plans['segmentation_export_params']['interpolation_order']
                interpolation_order_z = self.plans['segmentation_export_params']['interpolation_order_z']
            else:
                force_separate_z, interpolation_order, interpolation_order_z = None, 1, 0
        else:
            force_separate_z = segmentation_export_kwargs['force_separate_z']
            interpolation_order = segmentation_export_kwargs['interpolation_order']
            interpolation_order_z = segmentation_export_kwargs['interpolation_order_z']
        output_folder = join(self.output_folder, validation_folder_name)
        os.makedirs(output_folder, exist_ok=True)
        my_input_args = {'do_mirroring': do_mirroring, 'use_sliding_window': use_sliding_window, 'step_size': step_size,
                         'save_softmax': save_softmax, 'use_gaussian': use_gaussian, 'overwrite': overwrite,
                         'validation_folder_name': validation_folder_name, 'debug': debug, 'all_in_gpu': all_in_gpu,
                         'segmentation_export_kwargs': segmentation_export_kwargs}
        with open(join(output_folder, "validation_args.json"), 'w') as f:
            json.dump(my_input_args, f, sort_keys=True, indent=4)
        if do_mirroring:
            if not self.data_aug_params['do_mirror']:
                raise RuntimeError("We did not train with mirroring so you cannot do inference with mirroring enabled")
            mirror_axes = self.data_aug_params['mirror_axes']
        else:
            mirror_axes = ()
        try:
            import SimpleITK as sitk
        except ImportError:
            sitk = None

        def default_writer(seg, path, props):
            if sitk is None:
                np.save(path[:-7] + ".npy", seg)
                return
            img = sitk.GetImageFromArray(seg.astype(np.uint8))               # segmentation_export.py:144-148
            img.SetSpacing(props['itk_spacing'])
            img.SetOrigin(props['itk_origin'])
            img.SetDirection(props['itk_direction'])
            sitk.WriteImage(img, path)

        def default_gt_reader(path):
            if sitk is not None and isfile(path):
                return sitk.GetArrayFromImage(sitk.ReadImage(path))
            if isfile(path[:-7] + ".npy"):
                return np.load(path[:-7] + ".npy")
            return None
        writer = default_writer if writer is None else writer
        gt_reader = default_gt_reader if gt_reader is None else gt_reader

        net = self.network
        keep = net.keep_on_device
        cases, grids = [], set()
        try:
            with torch.no_grad():
                for k in self.dataset_val.keys():
                    entry = self.dataset[k]
                    if 'properties' in entry:
                        properties = entry['properties']
                    else:
                        with open(entry['properties_file'], 'rb') as f:
                            properties = pickle.load(f)
                    fname = properties['list_of_data_files'][0].split("/")[-1][:-12]
                    out_nii = join(output_folder, fname + ".nii.gz")
                    npy_case = entry['data_file'][:-4] + ".npy"
                    data = np.load(npy_case) if isfile(npy_case) else np.load(entry['data_file'])['data']
                    data = np.array(data)                                     # (a writable copy: memory-mapped / npz arrays)
                    print(k, data.shape)
                    data[-1][data[-1] == -1] = 0
                    net.keep_on_device = True
                    seg_grid, softmax = self.predict_preprocessed_data_return_seg_and_softmax(
                        data[:-1], do_mirroring=do_mirroring, mirror_axes=mirror_axes, use_sliding_window=use_sliding_window,
                        step_size=step_size, use_gaussian=use_gaussian, all_in_gpu=all_in_gpu, verbose=False,
                        mixed_precision=self.fp16)
                    net.keep_on_device = keep
                    tb = self.transpose_backward
                    seg = export_segmentation(softmax.contiguous(), properties, tb, self.regions_class_order, force_separate_z,
                                              interpolation_order, interpolation_order_z)
                    if save_softmax:
                        # (float16 like segmentation_export.py:109; the reference stores the volume on the case's own grid, here the
                        #  network's grid is stored: it is what an ensembling step on this engine reads back)
                        np.savez_compressed(join(output_folder, fname + ".npz"),
                                            softmax=softmax.permute(0, *[i + 1 for i in tb]).cpu().numpy().astype(np.float16))
                    writer(seg, out_nii, properties)
                    gt_path = join(self.gt_niftis_folder, fname + ".nii.gz") if self.gt_niftis_folder else None
                    gt = gt_reader(gt_path) if gt_path is not None else None
                    if gt is not None and tuple(gt.shape) == tuple(seg.shape):
                        cases.append((seg, np.asarray(gt), out_nii, gt_path))
                        grids.add("original")
                    else:
                        # no readable ground-truth volume: score on the network's grid against the preprocessed case's own labels
                        cases.append((seg_grid.cpu().numpy().astype(np.uint8), data[-1].astype(np.int16), out_nii, entry['data_file']))
                        grids.add("preprocessed")
        finally:
            net.keep_on_device = keep
        print("finished prediction")
        print("evaluation of raw predictions")
        task = self.dataset_directory.split("/")[-1] if self.dataset_directory else ""
        scores = aggregate_scores(cases, labels=list(range(self.num_classes)), json_output_file=join(output_folder, "summary.json"),
                                  json_name=self.experiment_name + " val tiled %s" % (str(use_sliding_window)),
                                  json_description="" if grids == {"original"} else
                                  "scored on the network's grid against the preprocessed labels (no readable ground-truth volume)",
                                  json_author="Fabian", json_task=task)
        if run_postprocessing_on_folds:
            self.print_to_log_file("validate: determine_postprocessing (connected-component search, e2enet/postprocessing) is outside "
                                   "the MI355X hot path and was skipped; run it with the reference package on %s" % output_folder)
        if self.gt_niftis_folder and os.path.isdir(self.gt_niftis_folder) and self.output_folder_base:
            gt_nifti_folder = join(self.output_folder_base, "gt_niftis")                 # reference :1455-1477
            os.makedirs(gt_nifti_folder, exist_ok=True)
            for f_ in sorted(os.listdir(self.gt_niftis_folder)):
                if f_.endswith(".nii.gz") or f_.endswith(".npy"):
                    shutil.copy(join(self.gt_niftis_folder, f_), gt_nifti_folder)
        self.network.train(current_mode)
        return scores

    # ------------------------------------------------------------------------------------------ out of scope
    def preprocess_patient(self, input_files):
        """reference :425-452"""
        _out_of_scope("preprocess_patient()", "preprocessing package (e2enet/preprocessing: crop, resample, normalise)")

    def preprocess_predict_nifti(self, *args, **kwargs):
        """reference :454-489"""
        _out_of_scope("preprocess_predict_nifti()", "preprocessing and NIfTI export (e2enet/preprocessing, "
                                                    "e2enet/inference/segmentation_export.py)")

    def load_dataset(self):
        """reference :585-586 + dataloading/dataset_loading.py:97-118: the preprocessed cases of the stage folder as an ordered dict of
        file names (the arrays stay on disk: DataLoader3D memory-maps <case>.npy or reads <case>.npz['data'])."""
        folder = self.folder_with_preprocessed_data
        ids = sorted({f[:-4] for f in os.listdir(folder) if f.endswith(".npz") or (f.endswith(".npy") and not f.endswith("_segs.npy"))})
        self.dataset = OrderedDict()
        for c in ids:
            e = OrderedDict()
            e['data_file'] = join(folder, "%s.npz" % c)
            e['properties_file'] = join(folder, "%s.pkl" % c)
            if len(ids) <= 1000 and isfile(e['properties_file']):          # (num_cases_properties_loading_threshold)
                with open(e['properties_file'], 'rb') as f:
                    e['properties'] = pickle.load(f)
            self.dataset[c] = e

    def do_split(self):
        """reference :588-651: fold 'all' trains and validates on every case; otherwise splits_final.pkl of the dataset directory
        (created as a seeded 5-fold split when absent -- sklearn's KFold(5, shuffle=True, random_state=12345), restated with the same
        RandomState permutation), or a seeded 80:20 split when the fold is not in the file."""
        if self.fold == "all":
            tr_keys = val_keys = list(self.dataset.keys())
        else:
            splits_file = join(self.dataset_directory, "splits_final.pkl")
            if not isfile(splits_file):
                keys = np.sort(list(self.dataset.keys()))
                # sklearn KFold(n_splits=5, shuffle=True, random_state=12345).split: indices shuffled once by
                # check_random_state(12345).shuffle, then cut into 5 consecutive folds (the first n % 5 one longer)
                idx = np.arange(len(keys))
                np.random.RandomState(12345).shuffle(idx)
                sizes = np.full(5, len(keys) // 5, dtype=int)
                sizes[:len(keys) % 5] += 1
                splits, start = [], 0
                for sz in sizes:
                    test = np.sort(idx[start:start + sz])
                    start += sz
                    train = np.setdiff1d(np.arange(len(keys)), test)
                    splits.append(OrderedDict(train=keys[train], val=keys[test]))
                with open(splits_file, 'wb') as f:
                    pickle.dump(splits, f)
            else:
                with open(splits_file, 'rb') as f:
                    splits = pickle.load(f)
            if self.fold < len(splits):
                tr_keys, val_keys = list(splits[self.fold]['train']), list(splits[self.fold]['val'])
            else:
                rnd = np.random.RandomState(seed=12345 + self.fold)
                keys = np.sort(list(self.dataset.keys()))
                idx_tr = rnd.choice(len(keys), int(len(keys) * 0.8), replace=False)
                idx_val = [i for i in range(len(keys)) if i not in idx_tr]
                tr_keys, val_keys = [keys[i] for i in idx_tr], [keys[i] for i in idx_val]
        tr_keys.sort()
        val_keys.sort()
        self.dataset_tr = OrderedDict((i, self.dataset[i]) for i in tr_keys)
        self.dataset_val = OrderedDict((i, self.dataset[i]) for i in val_keys)

    def get_basic_generators(self):
        """reference :735-754 (3D branch)."""
        from ..dataloading.dataset_loading import DataLoader3D
        self.load_dataset()
        self.do_split()
        dl_tr = DataLoader3D(self.dataset_tr, self.basic_generator_patch_size, self.patch_size, self.batch_size, False,
                             oversample_foreground_percent=self.oversample_foreground_percent, pad_mode="constant",
                             pad_sides=self.pad_all_sides, memmap_mode='r')
        dl_val = DataLoader3D(self.dataset_val, self.patch_size, self.patch_size, self.batch_size, False,
                              oversample_foreground_percent=self.oversample_foreground_percent, pad_mode="constant",
                              pad_sides=self.pad_all_sides, memmap_mode='r')
        return dl_tr, dl_val
