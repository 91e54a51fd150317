"""shiftConvPP network (``Generic_UNetPlusPlus``) on the MI355X engine.

Drop-in for reference e2enet/network_architecture/unetpp_d.py: same constructor signature (:227-236), same
attributes read by the trainer / inference code, same module tree and therefore the same ``state_dict`` names
(the checkpoint wire format), same parameter registration order (:418-438) and the same RNG draw order, so
``torch.manual_seed(s); Generic_UNetPlusPlus(...)`` yields bit-identical initial weights.

The nn modules below are parameter containers only.  ``forward`` runs the HIP engine (``engine.Engine``): depth
shift, concat, conv, InstanceNorm, LeakyReLU, transposed conv, pooling and heads are hand-written gfx950
kernels behind the C ABI of libe2e_hip.so.  There is no CPU / eager fallback: a CPU tensor raises.
"""
import os
import weakref
from copy import deepcopy

import numpy as np
import torch
from torch import nn

from .initialization import InitWeights_He          # noqa: F401  (re-exported like the reference module)
from .neural_network import SegmentationNetwork
from ..utilities.nd_softmax import softmax_helper
from ..engine import Engine, NetConfig


class torch_shift(nn.Module):
    """Restricted depth shift (reference unetpp_d.py:38-59).  On the engine the shift is an index offset in the
    conv load stage and is never materialised; the module exists for structural parity only."""

    def __init__(self, shift_size, dim, dim_num):
        super().__init__()
        self.shift_size, self.dim, self.dim_num = shift_size, dim, dim_num

    def forward(self, x):
        raise RuntimeError("torch_shift is fused into the convolution load stage of the HIP engine; "
                           "run the enclosing Generic_UNetPlusPlus instead")


class ConvDropoutNormNonlin(nn.Module):
    """Parameter container for shift -> conv(1,3,3) -> [dropout] -> InstanceNorm -> LeakyReLU
    (reference unetpp_d.py:61-111)."""

    def __init__(self, input_channels, output_channels, conv_op=nn.Conv3d, conv_kwargs=None, norm_op=nn.InstanceNorm3d,
                 norm_op_kwargs=None, dropout_op=nn.Dropout3d, dropout_op_kwargs=None, nonlin=nn.LeakyReLU,
                 nonlin_kwargs=None, shift_size=5):
        super().__init__()
        if nonlin_kwargs is None:
            nonlin_kwargs = {'negative_slope': 1e-2, 'inplace': True}
        if dropout_op_kwargs is None:
            dropout_op_kwargs = {'p': 0.5, 'inplace': True}
        if norm_op_kwargs is None:
            norm_op_kwargs = {'eps': 1e-5, 'affine': True, 'momentum': 0.1}
        if conv_kwargs is None:
            conv_kwargs = {'kernel_size': 3, 'stride': 1, 'padding': 1, 'dilation': 1, 'bias': True}
        self.nonlin_kwargs, self.nonlin = nonlin_kwargs, nonlin
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.norm_op_kwargs, self.conv_kwargs = norm_op_kwargs, conv_kwargs
        self.conv_op, self.norm_op = conv_op, norm_op
        self.shift_size = 5
        self.shift_D = torch_shift(self.shift_size, 2, 3)
        self.conv = self.conv_op(input_channels, output_channels, **self.conv_kwargs)
        if self.dropout_op is not None and self.dropout_op_kwargs['p'] is not None and self.dropout_op_kwargs['p'] > 0:
            self.dropout = self.dropout_op(**self.dropout_op_kwargs)
        else:
            self.dropout = None
        self.instnorm = self.norm_op(output_channels, **self.norm_op_kwargs)
        self.lrelu = self.nonlin(**self.nonlin_kwargs)

    def forward(self, x):
        raise RuntimeError("ConvDropoutNormNonlin blocks execute inside the HIP engine of Generic_UNetPlusPlus")


class StackedConvLayers(nn.Module):
    """reference unetpp_d.py:122-185"""

    def __init__(self, input_feature_channels, output_feature_channels, num_convs, conv_op=nn.Conv3d, conv_kwargs=None,
                 norm_op=nn.InstanceNorm3d, norm_op_kwargs=None, dropout_op=nn.Dropout3d, dropout_op_kwargs=None,
                 nonlin=nn.LeakyReLU, nonlin_kwargs=None, first_stride=None, basic_block=ConvDropoutNormNonlin):
        self.input_channels = input_feature_channels
        self.output_channels = output_feature_channels
        if conv_kwargs is None:
            conv_kwargs = {'kernel_size': 3, 'stride': 1, 'padding': 1, 'dilation': 1, 'bias': True}
        self.conv_kwargs = conv_kwargs
        if first_stride is not None:
            self.conv_kwargs_first_conv = deepcopy(conv_kwargs)
            self.conv_kwargs_first_conv['stride'] = first_stride
        else:
            self.conv_kwargs_first_conv = conv_kwargs
        super().__init__()
        common = (norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs)
        blocks = [basic_block(input_feature_channels, output_feature_channels, conv_op, self.conv_kwargs_first_conv, *common)]
        blocks += [basic_block(output_feature_channels, output_feature_channels, conv_op, self.conv_kwargs, *common)
                   for _ in range(num_convs - 1)]
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        return self.blocks(x)


_VARIANT_PERM = {"133": (0, 1, 2), "313": (1, 0, 2), "331": (2, 0, 1)}
_VARIANT_KERNEL = {"133": (1, 3, 3), "313": (3, 1, 3), "331": (3, 3, 1)}


class _EngineFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward/backward are the engine's op lists."""

    @staticmethod
    def forward(ctx, net, x, deep_supervision, *params):
        eng = net._engine_for(x)
        ctx.net, ctx.eng, ctx.ds = net, eng, deep_supervision
        outs = eng.forward(x, deep_supervision)
        ctx.generation = eng.generation
        outs = outs if isinstance(outs, list) else [outs]
        return tuple(o.clone() for o in outs)

    @staticmethod
    def backward(ctx, *grad_outputs):
        eng = ctx.eng
        if eng.generation != ctx.generation:
            # the engine keeps ONE set of activation buffers per (batch, patch) shape; a second forward has overwritten them
            raise RuntimeError("Generic_UNetPlusPlus: backward() after another forward() through the same network: the "
                               "engine reuses its activation buffers in place; call backward() before the next forward "
                               "(gradient accumulation over several forwards is not supported)")
        dl = list(grad_outputs) + [None] * (len(eng.heads) - len(grad_outputs))
        dl = [g.contiguous() if g is not None else None for g in dl]
        grads = eng.backward(dl)
        names = ctx.net._param_names
        return (None, None, None) + tuple(grads[n] for n in names)


class Generic_UNetPlusPlus(SegmentationNetwork):
    DEFAULT_BATCH_SIZE_3D = 2
    DEFAULT_PATCH_SIZE_3D = (64, 192, 160)
    SPACING_FACTOR_BETWEEN_STAGES = 2
    BASE_NUM_FEATURES_3D = 30
    MAX_NUMPOOL_3D = 999
    MAX_NUM_FILTERS_3D = 320

    def __init__(self, img_size, input_channels, base_num_features, num_classes, num_pool, num_conv_per_stage=2,
                 feat_map_mul_on_downscale=2, conv_op=nn.Conv3d, norm_op=nn.InstanceNorm3d, norm_op_kwargs=None,
                 dropout_op=nn.Dropout3d, dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None,
                 deep_supervision=True, dropout_in_localization=False, final_nonlin=softmax_helper,
                 weightInitializer=InitWeights_He(1e-2), pool_op_kernel_sizes=None, conv_kernel_sizes=None,
                 upscale_logits=False, convolutional_pooling=False, convolutional_upsampling=False,
                 max_num_features=None, basic_block=ConvDropoutNormNonlin, seg_output_use_bias=False, shift_size=5,
                 conv_variant="133"):
        super().__init__()
        # conv_variant: "133" = the shiftConvPP network; "313" / "331" = the reference's ablation networks
        # unetpp_d_313.py / unetpp_d_331.py (conv kernel (3,1,3) / (3,3,1), shift switched off in their source, :102).
        # They run on the SAME kernels: a (3,1,3) conv on [D,H,W] volumes is the (1,3,3) conv on the volumes stored as
        # [H,D,W], and a contiguous [o,i,3,1,3] weight tensor has the memory layout of [o,i,1,3,3] (kd in the row slot).
        # So the engine works on axis-permuted tensors (engine axis a = reference axis _perm[a]); only the transposed-conv
        # weights and the pooling plan have to be permuted, and forward() permutes at the boundary.
        if conv_variant not in _VARIANT_PERM:
            raise ValueError("conv_variant must be one of %s" % sorted(_VARIANT_PERM))
        self.conv_variant = conv_variant
        self._perm = _VARIANT_PERM[conv_variant]
        self._inv_perm = tuple(self._perm.index(a) for a in range(3))
        if conv_variant != "133":
            shift_size = 1
        # ---- what the engine supports: exactly the configuration nnUNetTrainer_simple builds (:292-301) ----
        if conv_op != nn.Conv3d:
            raise ValueError("the MI355X engine implements the 3D shiftConvPP network only (conv_op=nn.Conv3d)")
        if norm_op != nn.InstanceNorm3d or nonlin != nn.LeakyReLU:
            raise ValueError("engine supports InstanceNorm3d + LeakyReLU blocks (nnUNetTrainer_simple.py:276-279)")
        if not (convolutional_pooling and convolutional_upsampling):
            raise ValueError("shiftConvPP is built with convolutional_pooling=convolutional_upsampling=True "
                             "(nnUNetTrainer_simple.py:300-301)")
        if upscale_logits or seg_output_use_bias or feat_map_mul_on_downscale != 2:
            raise ValueError("unsupported: upscale_logits / seg_output_use_bias / feat_map_mul_on_downscale != 2")
        if num_pool != 5:
            # the reference's forward() indexes six levels literally and fails otherwise (unetpp_d.py:451-483)
            raise ValueError("shiftConvPP needs exactly 5 pooling stages")
        if nonlin_kwargs is None:
            nonlin_kwargs = {'negative_slope': 1e-2, 'inplace': True}
        if dropout_op_kwargs is None:
            dropout_op_kwargs = {'p': 0.5, 'inplace': True}
        if norm_op_kwargs is None:
            norm_op_kwargs = {'eps': 1e-5, 'affine': True, 'momentum': 0.1}
        if dropout_op_kwargs.get('p') not in (None, 0, 0.0):
            raise ValueError("dropout p > 0 is not supported by the engine (trainer uses p=0, nnUNetTrainer_simple.py:277)")
        if abs(nonlin_kwargs.get('negative_slope', 1e-2) - 1e-2) > 0 or abs(norm_op_kwargs.get('eps', 1e-5) - 1e-5) > 0 \
                or not norm_op_kwargs.get('affine', True):
            raise ValueError("engine is specialised for LeakyReLU(0.01) and InstanceNorm(eps=1e-5, affine=True)")
        self.convolutional_upsampling = convolutional_upsampling
        self.convolutional_pooling = convolutional_pooling
        self.upscale_logits = upscale_logits
        self.conv_kwargs = {'stride': 1, 'dilation': 1, 'bias': True}
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op_kwargs, self.norm_op_kwargs = dropout_op_kwargs, norm_op_kwargs
        self.weightInitializer = weightInitializer
        self.conv_op, self.norm_op, self.dropout_op = conv_op, norm_op, dropout_op
        self.num_classes = num_classes
        self.final_nonlin = final_nonlin
        self._deep_supervision = deep_supervision
        self.do_ds = deep_supervision

        if pool_op_kernel_sizes is None:
            pool_op_kernel_sizes = [(2, 2, 2)] * num_pool
        conv_kernel_sizes = [_VARIANT_KERNEL[conv_variant]] * (num_pool + 1)          # forced, reference :286-287
        # pooling plan in engine axis order (what the transposed convs, max-pools and strided convs of the modules below use)
        self._pool_e = [tuple(int(k[a]) for a in self._perm) for k in pool_op_kernel_sizes]
        self.input_shape_must_be_divisible_by = np.prod(pool_op_kernel_sizes, 0, dtype=np.int64)
        self.pool_op_kernel_sizes = pool_op_kernel_sizes
        self.conv_kernel_sizes = conv_kernel_sizes
        self.conv_pad_sizes = [[1 if i == 3 else 0 for i in k] for k in conv_kernel_sizes]
        self.max_num_features = self.MAX_NUM_FILTERS_3D if max_num_features is None else max_num_features

        blk = (self.conv_op, None, self.norm_op, self.norm_op_kwargs, self.dropout_op, self.dropout_op_kwargs,
               self.nonlin, self.nonlin_kwargs)

        def stacked(cin, cout, n, first_stride=None):
            kw = dict(self.conv_kwargs)
            kw['kernel_size'] = _VARIANT_KERNEL[conv_variant]
            kw['padding'] = [1 if v == 3 else 0 for v in kw['kernel_size']]
            return StackedConvLayers(cin, cout, n, blk[0], kw, *blk[2:], first_stride, basic_block=basic_block)

        # ---- construction order == reference (RNG parity): encoder, bottleneck, nests 0..4, heads ----
        context = []
        out_f, in_f = base_num_features, input_channels
        for d in range(num_pool):
            first_stride = pool_op_kernel_sizes[d - 1] if d != 0 else None
            context.append(stacked(in_f, out_f, num_conv_per_stage, first_stride))
            in_f = out_f
            out_f = min(int(np.round(out_f * feat_map_mul_on_downscale)), self.max_num_features)
        final_num_features = out_f
        context.append(nn.Sequential(stacked(in_f, out_f, num_conv_per_stage - 1, pool_op_kernel_sizes[-1]),
                                     stacked(out_f, final_num_features, 1)))
        self._context_out = [c.output_channels for c in context[:-1]] + [final_num_features]

        locs, ups, downs = [], [], []
        enc_feat = final_num_features
        for z in range(5):
            loc, up, enc_feat, down = self._create_nest(z, num_pool, enc_feat, num_conv_per_stage, stacked)
            locs.append(loc)
            ups.append(up)
            downs.append(down)
        heads = [conv_op(self._context_out[h], num_classes, 1, 1, 0, 1, 1, seg_output_use_bias) for h in range(4)]

        # ---- registration order == reference (:418-438): parameter order, Masking draw order ----
        for z in range(5):
            setattr(self, "loc%d" % z, nn.ModuleList(locs[z]))
        self.conv_blocks_context = nn.ModuleList(context)
        self.td = nn.ModuleList([])
        for z in range(5):
            setattr(self, "up%d" % z, nn.ModuleList(ups[z]))
        for z in range(5):
            setattr(self, "down%d" % z, nn.ModuleList(downs[z]))
        self.seg_outputs = nn.ModuleList(heads)
        self.upscale_logits_ops = [lambda x: x for _ in range(num_pool - 1)]

        if self.weightInitializer is not None:
            self.apply(self.weightInitializer)

        # shift_size: keyword extension (default = the reference's hard-set 5, unetpp_d.py:89); 3/7/11 are the sizes its
        # comment lists, 1 reproduces the 'noshift' ablation variant
        if conv_variant != "133":
            # He-init parity: the reference draws a transposed-conv weight as [in, out, kD, kH, kW]; the same draws, taken
            # in that shape and moved to the engine's axis order, are the same network
            with torch.no_grad():
                for n, p_ in self.named_parameters():
                    if self._is_up_weight(n):
                        p_.copy_(self._up_to_engine(p_.detach().clone().view(self._up_ref_shape(p_))))
            self._register_state_dict_hook(Generic_UNetPlusPlus._state_dict_to_reference)
            self._register_load_state_dict_pre_hook(self._state_dict_from_reference)
        self._cfg = NetConfig(input_channels, base_num_features, num_classes, self._pool_e, num_conv_per_stage,
                              self.max_num_features, shift_size=shift_size)
        self._engines = {}
        self._kernel_masks = None            # name -> uint8 [dim0, dim1]; None = dense
        self._auto_sparsity = False          # derive liveness from zero kernels (inference on DSFF checkpoints)
        self._weights_outside_masks = False  # weights were (re)loaded after the masks were pushed: see _on_state_loaded
        self._param_names = [n for n, _ in self.named_parameters()]
        self.register_load_state_dict_post_hook(lambda module, keys: module._on_state_loaded())

    # ---- axis-permuted variants: layout helpers --------------------------------------------------------------------
    @staticmethod
    def _is_up_weight(name):
        return name.startswith("up") and name.endswith(".weight")

    def _up_ref_shape(self, w):
        """shape of an engine-order transposed-conv weight in the reference's axis order"""
        k = tuple(w.shape[2:])
        return tuple(w.shape[:2]) + tuple(k[self._inv_perm[a]] for a in range(3))

    def _up_to_engine(self, w_ref):
        return w_ref.permute(0, 1, *[2 + a for a in self._perm]).contiguous()

    def _up_to_reference(self, w_eng):
        return w_eng.permute(0, 1, *[2 + a for a in self._inv_perm]).contiguous()

    @staticmethod
    def _state_dict_to_reference(module, state_dict, prefix, local_metadata):
        """state_dict(): tensors in the reference's shapes (checkpoint wire format)"""
        for key in list(state_dict.keys()):
            name = key[len(prefix):]
            if module._is_up_weight(name):
                state_dict[key] = module._up_to_reference(state_dict[key])
        return state_dict

    def _state_dict_from_reference(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for key in list(state_dict.keys()):
            if key.startswith(prefix) and self._is_up_weight(key[len(prefix):]):
                state_dict[key] = self._up_to_engine(state_dict[key])

    def to_engine_layout(self, t: torch.Tensor) -> torch.Tensor:
        """[N, C, D, H, W] tensor in the reference's axis order -> the axis order the engine of this network runs in
        (identity for the (1,3,3) network).  The trainer's fast path feeds ``engine()`` with data and targets in this layout."""
        if self.conv_variant == "133":
            return t
        return t.permute(0, 1, *[2 + a for a in self._perm]).contiguous()

    def from_engine_layout(self, t: torch.Tensor) -> torch.Tensor:
        if self.conv_variant == "133":
            return t
        return t.permute(0, 1, *[2 + a for a in self._inv_perm]).contiguous()

    def _create_nest(self, z, num_pool, final_num_features, n_conv, stacked):
        """reference create_nest (:491-550) for convolutional_upsampling=True."""
        loc, tu, tdown = [], [], []
        unet_final = None
        for u in range(z, num_pool):
            from_down = final_num_features
            from_skip = self._context_out[-(2 + u)]
            concat = from_skip * 2 + (self._context_out[-(3 + u)] if u != num_pool - 1 else 0)
            if unet_final is None:
                unet_final = from_skip
            final_num_features = from_skip
            # (engine axis order: the weight tensor [in, out, k0, k1, k2] is what the kernels read)
            tu.append(nn.ConvTranspose3d(from_down, from_skip, self._pool_e[-(u + 1)], self._pool_e[-(u + 1)], bias=False))
            if u + 2 <= len(self._pool_e):
                tdown.append(nn.MaxPool3d(self._pool_e[-(u + 2)]))
            if z != 0:
                loc.append(nn.Sequential(stacked(concat, final_num_features, n_conv - 1)))
            else:
                loc.append(nn.Sequential(stacked(concat, from_skip, n_conv - 1), stacked(from_skip, final_num_features, 1)))
        return loc, tu, unet_final, tdown

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _live_params(self):
        return dict(self.named_parameters())

    def _engine_for(self, x: torch.Tensor) -> Engine:
        key = (tuple(x.shape), x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            div = [int(self.input_shape_must_be_divisible_by[a]) for a in self._perm]      # x is in engine axis order
            if any(int(s) % int(d) for s, d in zip(x.shape[2:], div)):
                raise ValueError("input spatial shape %s must be divisible by %s" % (tuple(x.shape[2:]), tuple(div)))
            eng = Engine(self._cfg, self._live_params(), x.shape[0], tuple(x.shape[2:]), x.device)
            eng._sparsity_version = -1
            ref = weakref.ref(self)

            def _refresh(eng_ref=weakref.ref(eng)):
                net, e = ref(), eng_ref()
                if net is not None and e is not None:
                    net._sync_sparsity(e)          # (a version compare unless the masks changed)
            eng.pre_forward_hook = _refresh
            self._engines[key] = eng
            # plans are kept per (batch, patch) shape, least recently used first out, inside a byte budget (288 GB of HBM:
            # a training plan at 2 x 128^3 holds 8.6 GB, the 8-mirror inference plan 17 GB; alternating between them must
            # not re-allocate every time)
            budget = float(os.environ.get("E2E_PLAN_CACHE_GB", "96")) * 2 ** 30
            while len(self._engines) > 1 and sum(e.activation_bytes() for e in self._engines.values()) > budget:
                self._engines.pop(next(iter(self._engines)))
        else:
            self._engines[key] = self._engines.pop(key)     # most recently used last
        eng.params = self._live_params()
        self._sync_sparsity(eng)
        return eng

    def weights_changed(self):
        """Parameters were modified in a way the plans cannot see -- in place through ``parameter.data`` (a view with its own
        version counter) or by foreign native code: the packed matrix-pipe weights and operand ranges are rebuilt at the next pass.
        Not needed after optimizer steps, ``load_state_dict``, ``Masking`` updates or any in-place op on the parameters themselves."""
        for eng in self._engines.values():
            eng.weights_changed()

    def _invalidate_sparsity(self):
        self._sparsity_version = getattr(self, "_sparsity_version", 0) + 1

    def _on_state_loaded(self):
        """load_state_dict ran.  The reference creates ``Masking`` (fresh random masks) BEFORE it loads a checkpoint
        (simple_main.py:163-177): until the next ``apply_mask`` the loaded weights are used as they are, whatever the
        masks say (the reference's conv is dense).  So from here to the next mask application a kernel is alive when
        its mask says so OR when it holds a non-zero weight."""
        self._weights_outside_masks = self._kernel_masks is not None
        self._invalidate_sparsity()

    def masks_applied(self):
        """Called by ``Masking`` after weights *= mask (apply_mask or the fused optimizer step): masks are authoritative."""
        if self._weights_outside_masks:
            self._weights_outside_masks = False
            self._invalidate_sparsity()

    def set_kernel_masks(self, kmasks):
        """DSFF: name -> uint8 [dim0, dim1] kernel liveness map (from ``Masking``).  None = dense."""
        self._kernel_masks = kmasks
        self._auto_sparsity = False
        self._invalidate_sparsity()

    def enable_auto_sparsity(self, flag=True):
        """Inference on a DSFF checkpoint: pruned kernels are exact zeros, skip them.  Takes precedence over masks pushed
        by a ``Masking`` object (the reference's inference path never consults the masks)."""
        self._auto_sparsity = flag
        if flag:
            self._kernel_masks = None
            self._weights_outside_masks = False
        self._invalidate_sparsity()

    def _sync_sparsity(self, eng):
        ver = getattr(self, "_sparsity_version", 0)
        if eng._sparsity_version == ver:
            return
        if self._kernel_masks is not None:
            km = self._kernel_masks
            if self._weights_outside_masks:
                nz = eng.kernel_masks_from_weights(names=set(km.keys()))
                km = {n: (torch.bitwise_or(km[n].to(nz[n].device), nz[n]) if n in nz else km[n]) for n in km}
            eng.set_kernel_masks(km)
        elif self._auto_sparsity:
            eng.set_kernel_masks(eng.kernel_masks_from_weights())
        else:
            eng.set_kernel_masks({})
        eng._sparsity_version = ver

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("Generic_UNetPlusPlus (MI355X engine) needs a GPU tensor: there is no CPU fallback")
        x = self.to_engine_layout(x.float()).contiguous()
        ds = bool(self._deep_supervision and self.do_ds)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            params = [p for _, p in self.named_parameters()]
            outs = list(_EngineFunction.apply(self, x, ds, *params))
        else:
            eng = self._engine_for(x)
            o = eng.forward(x, ds)
            outs = [t.clone() for t in (o if isinstance(o, list) else [o])]
        outs = [self.final_nonlin(self.from_engine_layout(o)) for o in outs]
        return outs if ds else outs[0]

    def engine(self, x):
        """The execution plan for inputs shaped like ``x`` (fast path used by the trainer and the benchmark).  ``x`` is in
        engine axis order (``to_engine_layout``; the reference's own order for the (1,3,3) network)."""
        return self._engine_for(x)
