"""``nnUNetTrainer_simple`` on the MI355X engine.

Keeps the operator surface of reference e2enet/training/network_training/nnUNetTrainer_simple.py that
``simple_main.py`` / ``simple_predict.py`` / ``model_restore`` touch: constructor (:59-61), ``initialize`` (:178-253),
``initialize_network`` (:255-363, shiftConvPP branch :292-301), ``initialize_optimizer_and_scheduler`` (:367-371),
``run_iteration`` (:529-583), ``run_training`` (:929-1027), ``process_plans`` (:1036-1103), checkpoints (:1140-1255),
``predict_preprocessed_data_return_seg_and_softmax`` (:491-527).  The arithmetic of an iteration -- forward, deep
supervision Dice+CE, backward, clip_grad_norm_(12), SGD-Nesterov, DSFF mask -- is one chain of HIP kernel launches
(``engine.Engine`` + ``fused_optim.FusedClipSGD``), fp32 throughout.

Out of scope here: the batchgenerators data pipeline, NIfTI export and ``validate``.  Data
arrives through any iterator of ``{'data': [B,C,...], 'target': [list of [B,1,...] per scale]}`` dicts (the format
of the reference's augmenter output, :538-540); ``SyntheticGenerator`` provides seeded synthetic batches.
"""
import os
import pickle
from collections import OrderedDict
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


class nnUNetTrainer_simple(object):
    def __init__(self, plans_file, fold, output_folder=None, dataset_directory=None, batch_dice=True, stage=None,
                 unpack_data=True, deterministic=True, fp16=False, Tconv=None, max_num_epochs=200,
                 num_batches_per_epoch=100, args=None):
        self.init_args = (plans_file, fold, output_folder, dataset_directory, batch_dice, stage, unpack_data,
                          deterministic, fp16)
        self.plans_file, self.fold, self.output_folder = plans_file, fold, output_folder
        self.dataset_directory, self.batch_dice, self.stage = dataset_directory, batch_dice, stage
        self.unpack_data, self.deterministic = unpack_data, deterministic
        # the engine computes in fp32 (1e-4 logit parity bar); the reference's AMP switch is accepted and ignored
        self.fp16 = False
        self.Tconv = Tconv if Tconv is not None else 'shiftConvPP'
        self.args = args
        self.plans = None
        self.network = self.optimizer = self.lr_scheduler = None
        self.tr_gen = self.val_gen = None
        self.was_initialized = False
        self.loss = DC_and_CE_loss({'batch_dice': self.batch_dice, 'smooth': 1e-5, 'do_bg': False}, {})
        self.initial_lr, self.weight_decay = 1e-2, 3e-5
        self.max_num_epochs, self.num_batches_per_epoch = max_num_epochs, num_batches_per_epoch
        self.num_val_batches_per_epoch = 50
        self.save_every = 50
        self.epoch = 0
        self.deep_supervision_scales = self.ds_loss_weights = None
        self.all_tr_losses, self.all_val_losses, self.all_val_losses_tr_mode, self.all_val_eval_metrics = [], [], [], []
        self.best_epoch_based_on_MA_tr_loss = self.best_MA_tr_loss_for_patience = self.best_val_eval_criterion_MA = None
        self.regions_class_order = None
        self.data_aug_params = {'mirror_axes': (0, 1, 2), 'do_mirror': True}
        self.base_num_features_override = None      # reference hard-codes 48 for shiftConvPP (:296)
        self._fused = None
        self.log_file = None

    # ------------------------------------------------------------------------------------------ plans
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
        self.base_num_features = plans['base_num_features']
        self.num_input_channels = plans['num_modalities']
        self.num_classes = plans['num_classes'] + 1
        self.transpose_forward = plans.get('transpose_forward', [0, 1, 2])
        self.transpose_backward = plans.get('transpose_backward', [0, 1, 2])
        self.conv_per_stage = plans.get('conv_per_stage', 2)
        self.threeD = len(self.patch_size) == 3
        if not self.threeD:
            raise RuntimeError("the MI355X engine implements 3d_fullres plans only")

    def setup_DA_params(self):
        """reference :682-733: deep-supervision target scales from the cumulative pooling."""
        self.deep_supervision_scales = [[1, 1, 1]] + list(list(i) for i in 1 / np.cumprod(
            np.vstack(self.net_num_pool_op_kernel_sizes), axis=0))[:-1]

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
                scales = self.deep_supervision_scales[:4]
                dev = "cuda" if torch.cuda.is_available() else "cpu"
                self.tr_gen = SyntheticGenerator(self.batch_size, self.num_input_channels, self.patch_size,
                                                 self.num_classes, scales, seed=0, device=dev)
                self.val_gen = SyntheticGenerator(self.batch_size, self.num_input_channels, self.patch_size,
                                                  self.num_classes, scales, seed=1, device=dev)
            assert isinstance(self.network, (SegmentationNetwork, nn.DataParallel))
        self.was_initialized = True
        return self.network, self.optimizer

    def initialize_network(self):
        """reference :292-301 (Tconv == 'shiftConvPP'); other Tconv values are ablations outside this engine."""
        if self.Tconv != 'shiftConvPP':
            raise NotImplementedError("the MI355X engine implements Tconv='shiftConvPP' (got %r)" % (self.Tconv,))
        base = 48 if self.base_num_features_override is None else self.base_num_features_override
        self.network = Generic_UNetPlusPlus(self.patch_size, self.num_input_channels, base, self.num_classes,
                                            len(self.net_num_pool_op_kernel_sizes), self.conv_per_stage, 2, nn.Conv3d,
                                            nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, nn.Dropout3d,
                                            {'p': 0, 'inplace': True}, nn.LeakyReLU,
                                            {'negative_slope': 1e-2, 'inplace': True}, True, False, lambda x: x,
                                            InitWeights_He(1e-2), self.net_num_pool_op_kernel_sizes,
                                            self.net_conv_kernel_sizes, False, True, True)
        if torch.cuda.is_available():
            self.network.cuda()
        self.network.inference_apply_nonlin = softmax_helper

    def initialize_optimizer_and_scheduler(self):
        assert self.network is not None, "self.initialize_network must be called first"
        self.optimizer = torch.optim.SGD(self.network.parameters(), self.initial_lr, weight_decay=self.weight_decay,
                                         momentum=0.99, nesterov=True)
        self.lr_scheduler = None
        self._fused = None

    # ------------------------------------------------------------------------------------------ one iteration
    def run_iteration(self, data_generator, do_backprop=True, run_online_evaluation=False, mask=None):
        """reference :529-583.  Returns the loss as a numpy scalar (one device->host sync, like the reference)."""
        data_dict = next(data_generator)
        data, target = data_dict['data'], data_dict['target']
        dev = next(self.network.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("nnUNetTrainer_simple (MI355X) needs the network on a GPU: there is no CPU fallback")
        data = torch.as_tensor(data).float().to(dev, non_blocking=True).contiguous()
        if not isinstance(target, (list, tuple)):
            target = [target]
        target = [torch.as_tensor(t).float().to(dev, non_blocking=True).contiguous() for t in target]
        eng = self.network.engine(data)
        eng.forward(data, deep_supervision=True)
        if do_backprop:
            loss = eng.loss_backward(target, self.ds_loss_weights, batch_dice=self.batch_dice)
            if self._fused is None:
                self._fused = FusedClipSGD(self.optimizer, list(self.network.named_parameters()), max_norm=12.0)
            masks = mask.masks if mask is not None else None
            self._fused.step(eng.grads, masks)                  # clip + SGD + (weight, momentum) *= mask
            if mask is not None:
                mask.step(masks_already_applied=True)
        else:
            outs = [h.out.data for h in eng.heads]
            loss = self.loss(outs, target)
        return loss.detach().cpu().numpy().reshape(())

    def maybe_update_lr(self, epoch=None):
        ep = self.epoch + 1 if epoch is None else epoch
        self.optimizer.param_groups[0]['lr'] = poly_lr(ep, self.max_num_epochs, self.initial_lr, 0.9)

    def run_training(self, mask=None):
        """reference :929-1027 (epoch loop, poly LR, periodic checkpoints)."""
        self.maybe_update_lr(self.epoch)
        self.network.do_ds = True
        while self.epoch < self.max_num_epochs:
            t0 = time()
            self.network.train()
            losses = [self.run_iteration(self.tr_gen, True, mask=mask) for _ in range(self.num_batches_per_epoch)]
            self.all_tr_losses.append(float(np.mean(losses)))
            with torch.no_grad():
                self.network.eval()
                vl = [self.run_iteration(self.val_gen, False, True) for _ in range(min(self.num_val_batches_per_epoch, 2))]
                self.all_val_losses.append(float(np.mean(vl)))
            self.print_to_log_file("epoch %d: train loss %.4f val loss %.4f (%.1f s)" %
                                   (self.epoch, self.all_tr_losses[-1], self.all_val_losses[-1], time() - t0))
            self.maybe_update_lr()
            if self.output_folder and (self.epoch % self.save_every == self.save_every - 1):
                self.save_checkpoint(os.path.join(self.output_folder, self.Tconv + "_model_latest.model"))
            self.epoch += 1
        if self.output_folder:
            self.save_checkpoint(os.path.join(self.output_folder, self.Tconv + "_model_final_checkpoint.model"))

    def print_to_log_file(self, *args):
        print(*args)

    # ------------------------------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, fname, save_optimizer=True, mask=None):
        """reference :1140-1176 (same dict keys and the side-car .pkl).  With ``mask`` (a Masking) the checkpoint also
        carries 'dsff_state' (packed kernel maps, death-rate schedule position, growth RNG state); reference loaders
        ignore the extra key."""
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
        with open(fname + ".pkl", 'wb') as f:
            pickle.dump(info, f)

    def load_checkpoint_ram(self, checkpoint, train=True, mask=None):
        """reference :1211-1255; ``mask``: a Masking already attached to this trainer's network/optimizer
        (add_module done) that is restored from checkpoint['dsff_state']."""
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
        if mask is not None:
            if 'dsff_state' not in checkpoint:
                raise KeyError("checkpoint carries no 'dsff_state' (written by save_checkpoint(..., mask=mask))")
            mask.load_state_dict(checkpoint['dsff_state'])
        # DSFF checkpoints carry their masks implicitly (pruned kernels are exact zeros): skip them at inference
        if not train:
            self.network.enable_auto_sparsity(True)

    def load_checkpoint(self, fname, train=True, mask=None):
        self.load_checkpoint_ram(torch.load(fname, map_location=torch.device('cpu'), weights_only=False), train, mask)

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
