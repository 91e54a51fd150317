"""The 'shiftConvPP_nodff' ablation network on the MI355X engine.

Drop-in for reference e2enet/network_architecture/unetpp_d_nodff.py (:171-378), selected by
``nnUNetTrainer_simple`` for ``Tconv == 'shiftConvPP_nodff'`` (nnUNetTrainer_simple.py:326-335): the same shift-conv blocks
(depth shift of size 3, the block's default there, :55) wired as a plain U-Net -- encoder with strided first convolutions, per
level ``ConvTranspose3d`` -> ``cat((up, skip))`` -> two conv blocks -> 1x1x1 head -- without the nested dense feature fusion.
Same constructor signature, module tree / ``state_dict`` names (``conv_blocks_context``, ``conv_blocks_localization``, ``tu``,
``seg_outputs``), construction order (RNG parity of the He init) and ``num_pool`` deep-supervision outputs
``[full res, 1/2, ..., lowest]``.

Everything runs on the kernels of the shiftConvPP engine (``engine.Engine`` with the ``'unet'`` plan builder); the engine
plumbing (plans per shape, DSFF kernel maps, autograd node, sliding-window inference) is inherited from
``unetpp_d.Generic_UNetPlusPlus``.
"""
import numpy as np
import torch
from torch import nn

from . import unetpp_d as _base
from .initialization import InitWeights_He          # noqa: F401  (re-exported like the reference module)
from .neural_network import SegmentationNetwork
from ..utilities.nd_softmax import softmax_helper
from ..engine import NetConfig

torch_shift = _base.torch_shift
ConvDropoutNormNonlin = _base.ConvDropoutNormNonlin
StackedConvLayers = _base.StackedConvLayers


class Generic_UNetPlusPlus(_base.Generic_UNetPlusPlus):
    MAX_NUM_FILTERS_3D = 320

    def __init__(self, img_size, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2,
                 feat_map_mul_on_downscale=2, conv_op=nn.Conv3d, norm_op=nn.InstanceNorm3d, norm_op_kwargs=None,
                 dropout_op=nn.Dropout3d, dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None,
                 deep_supervision=True, dropout_in_localization=False, final_nonlin=softmax_helper,
                 weightInitializer=InitWeights_He(1e-2), pool_op_kernel_sizes=None, conv_kernel_sizes=None,
                 upscale_logits=False, convolutional_pooling=False, convolutional_upsampling=False,
                 max_num_features=None, basic_block=ConvDropoutNormNonlin, seg_output_use_bias=False, shift_size=3):
        SegmentationNetwork.__init__(self)
        self.conv_variant = "133"
        self._perm = self._inv_perm = (0, 1, 2)
        # ---- what the engine supports: the configuration nnUNetTrainer_simple builds (:326-335) ----
        if conv_op != nn.Conv3d or norm_op != nn.InstanceNorm3d or nonlin != nn.LeakyReLU:
            raise ValueError("the MI355X engine implements Conv3d + InstanceNorm3d + LeakyReLU blocks only")
        if not (convolutional_pooling and convolutional_upsampling):
            raise ValueError("shiftConvPP_nodff is built with convolutional_pooling=convolutional_upsampling=True "
                             "(nnUNetTrainer_simple.py:335)")
        if upscale_logits or seg_output_use_bias or feat_map_mul_on_downscale != 2:
            raise ValueError("unsupported: upscale_logits / seg_output_use_bias / feat_map_mul_on_downscale != 2")
        if nonlin_kwargs is None:
            nonlin_kwargs = {'negative_slope': 1e-2, 'inplace': True}
        if dropout_op_kwargs is None:
            dropout_op_kwargs = {'p': 0.5, 'inplace': True}
        if norm_op_kwargs is None:
            norm_op_kwargs = {'eps': 1e-5, 'affine': True, 'momentum': 0.1}
        if dropout_op_kwargs.get('p') not in (None, 0, 0.0):
            raise ValueError("dropout p > 0 is not supported by the engine (trainer uses p=0)")
        if abs(nonlin_kwargs.get('negative_slope', 1e-2) - 1e-2) > 0 or abs(norm_op_kwargs.get('eps', 1e-5) - 1e-5) > 0 \
                or not norm_op_kwargs.get('affine', True):
            raise ValueError("engine is specialised for LeakyReLU(0.01) and InstanceNorm(eps=1e-5, affine=True)")
        self.convolutional_upsampling = convolutional_upsampling
        self.convolutional_pooling = convolutional_pooling
        self.upscale_logits = upscale_logits
        self.conv_kwargs = {'stride': 1, 'dilation': 1, 'bias': True}
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.norm_op, self.norm_op_kwargs = norm_op, norm_op_kwargs
        self.conv_op = conv_op
        self.weightInitializer = weightInitializer
        self.num_classes = num_classes
        self.final_nonlin = final_nonlin
        self._deep_supervision = deep_supervision
        self.do_ds = deep_supervision

        if pool_op_kernel_sizes is None:
            pool_op_kernel_sizes = [(2, 2, 2)] * num_pool
        assert len(pool_op_kernel_sizes) == num_pool
        conv_kernel_sizes = [(1, 3, 3)] * (num_pool + 1)                    # forced, reference :222-223
        self._pool_e = [tuple(int(v) for v in k) for k in pool_op_kernel_sizes]
        self.input_shape_must_be_divisible_by = np.prod(pool_op_kernel_sizes, 0, dtype=np.int64)
        self.pool_op_kernel_sizes = pool_op_kernel_sizes
        self.conv_kernel_sizes = conv_kernel_sizes
        self.conv_pad_sizes = [[i // 2 for i in k] for k in conv_kernel_sizes]
        self.max_num_features = self.MAX_NUM_FILTERS_3D if max_num_features is None else max_num_features

        common = (self.norm_op, self.norm_op_kwargs, self.dropout_op, self.dropout_op_kwargs, self.nonlin, self.nonlin_kwargs)

        def stacked(cin, cout, n, first_stride=None):
            kw = dict(self.conv_kwargs)
            kw['kernel_size'] = (1, 3, 3)
            kw['padding'] = [0, 1, 1]
            return StackedConvLayers(cin, cout, n, self.conv_op, kw, *common, first_stride, basic_block=basic_block)

        # ---- module registration and construction order == reference (:238-316): RNG parity of the init ----
        self.conv_blocks_context = nn.ModuleList()
        self.conv_blocks_localization = nn.ModuleList()
        self.td = nn.ModuleList()
        self.tu = nn.ModuleList()
        self.seg_outputs = nn.ModuleList()
        in_f, out_f = input_channels, base_num_features
        for d in range(num_pool):
            first_stride = pool_op_kernel_sizes[d - 1] if d != 0 else None
            self.conv_blocks_context.append(stacked(in_f, out_f, num_conv_per_stage, first_stride))
            in_f = out_f
            out_f = min(int(np.round(out_f * feat_map_mul_on_downscale)), self.max_num_features)
        final_num_features = out_f
        self.conv_blocks_context.append(nn.Sequential(stacked(in_f, out_f, num_conv_per_stage - 1, pool_op_kernel_sizes[-1]),
                                                      stacked(out_f, final_num_features, 1)))
        for u in range(num_pool):
            from_down = final_num_features
            from_skip = self.conv_blocks_context[-(2 + u)].output_channels
            final_num_features = from_skip
            self.tu.append(nn.ConvTranspose3d(from_down, from_skip, pool_op_kernel_sizes[-(u + 1)],
                                              pool_op_kernel_sizes[-(u + 1)], bias=False))
            self.conv_blocks_localization.append(nn.Sequential(stacked(from_skip * 2, from_skip, num_conv_per_stage - 1),
                                                               stacked(from_skip, final_num_features, 1)))
        for ds in range(len(self.conv_blocks_localization)):
            self.seg_outputs.append(conv_op(self.conv_blocks_localization[ds][-1].output_channels, num_classes, 1, 1, 0, 1, 1,
                                            seg_output_use_bias))
        self.upscale_logits_ops = [lambda x: x for _ in range(num_pool - 1)]
        if self.weightInitializer is not None:
            self.apply(self.weightInitializer)

        self._cfg = NetConfig(input_channels, base_num_features, num_classes, self._pool_e, num_conv_per_stage,
                              self.max_num_features, shift_size=shift_size, graph="unet")
        self._engines = {}
        self._kernel_masks = None
        self._auto_sparsity = False
        self._weights_outside_masks = False
        self._param_names = [n for n, _ in self.named_parameters()]
        self.register_load_state_dict_post_hook(lambda module, keys: module._on_state_loaded())

    @staticmethod
    def _is_up_weight(name):
        return name.startswith("tu.") and name.endswith(".weight")
