"""``SegmentationNetwork``: sliding-window / mirrored inference with the aggregation kept in HBM.

Drop-in for reference e2enet/network_architecture/neural_network.py (``predict_3D`` :72-162 and helpers).  Same
signature and return values ``(seg int64 [X,Y,Z], probs float32 [K,X,Y,Z])``.  Differences in mechanism only:
  * tiles, mirror flips, softmax, Gaussian weighting, overlap-add, normalisation and argmax run as HIP kernels on
    device buffers (the reference copies every tile to the host and adds in numpy, :390-393);
  * tiles can be sharded over the ranks of a ``torch.distributed`` group (RCCL over xGMI): each rank evaluates
    its tiles, one all-gather exchanges the Gaussian-weighted probability patches, and every rank overlap-adds
    them in the reference's x->y->z order, so the result is bit-identical to the single-GPU run.
Inference runs in fp32 (the reference's ``mixed_precision`` flag is accepted and ignored: the 1e-4 logit parity
bar needs fp32).
"""
from typing import List, Optional, Tuple, Union

import os

import numpy as np
import torch
from torch import nn
from scipy.ndimage import gaussian_filter

from .._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def pad_nd_image(image: np.ndarray, new_shape, mode="constant", kwargs=None, return_slicer=False,
                 shape_must_be_divisible_by=None):
    """Symmetric padding of the trailing axes up to ``new_shape`` (semantics of batchgenerators==0.24
    ``pad_nd_image``, the third-party helper the reference imports at neural_network.py:17)."""
    if kwargs is None:
        kwargs = {'constant_values': 0}
    old = np.array(image.shape[-len(new_shape):])
    lead = image.ndim - len(new_shape)
    target = np.array([max(int(new_shape[i]), int(old[i])) for i in range(len(new_shape))])
    if shape_must_be_divisible_by is not None:
        div = np.array(shape_must_be_divisible_by).reshape(-1)
        if div.size == 1:
            div = np.repeat(div, len(target))
        for i in range(len(target)):
            if target[i] % div[i] != 0:
                target[i] += div[i] - target[i] % div[i]
    diff = target - old
    below, above = diff // 2, diff // 2 + diff % 2
    pads = [[0, 0]] * lead + [[int(a), int(b)] for a, b in zip(below, above)]
    res = np.pad(image, pads, mode, **kwargs) if diff.any() else image
    if not return_slicer:
        return res
    pads = np.array(pads)
    pads[:, 1] = np.array(res.shape) - pads[:, 1]
    return res, [slice(int(a), int(b)) for a, b in pads]


class NeuralNetwork(nn.Module):
    def __init__(self):
        super().__init__()

    def get_device(self):
        dev = next(self.parameters()).device
        return "cpu" if dev.type == "cpu" else dev.index

    def set_device(self, device):
        if device == "cpu":
            self.cpu()
        else:
            self.cuda(device)

    def forward(self, x):
        raise NotImplementedError


class SegmentationNetwork(NeuralNetwork):
    def __init__(self):
        super().__init__()
        self.input_shape_must_be_divisible_by = None
        self.conv_op = None
        self.num_classes = None
        self.inference_apply_nonlin = lambda x: x
        self._gaussian_3d = self._patch_size_for_gaussian_3d = None
        self._gaussian_3d_dev = None
        # tile sharding over a process group (None = this process evaluates every tile)
        self.tile_group = None
        self.tile_rank, self.tile_world = 0, 1
        self.tile_force = False
        # how the ranks' tiles meet: "allgather" = per group of `world` tiles one all-gather of the probability
        # patches, every rank overlap-adds all tiles in the reference's order (bit-identical to one process); "allreduce" = every
        # rank overlap-adds its own tiles into a partial volume, one all-reduce at the end (fewer bytes, <= 1e-6 from one process)
        self.tile_exchange = "allgather"
        # inference/predict.py (fold ensembling + export on device): predict_3D then returns device tensors
        self.keep_on_device = False

    # ------------------------------------------------------------------------------------------ configuration
    def shard_tiles(self, rank: int, world: int, group=None, force: bool = False, exchange: str = None):
        """Evaluate tiles ``rank::world`` of the x->y->z tile list on this process and exchange the weighted
        probability patches with one all-gather over ``group`` (RCCL on GPUs, gloo in CPU tests).  ``force`` takes the
        sharded branch (all-gather included) even for a single rank: self-test of the exchange on one GPU.
        ``exchange``: "allgather" (default) or "allreduce" (see ``tile_exchange``)."""
        self.tile_rank, self.tile_world, self.tile_group, self.tile_force = int(rank), int(world), group, bool(force)
        if exchange is not None:
            if exchange not in ("allgather", "allreduce"):
                raise ValueError("exchange must be 'allgather' or 'allreduce', got %r" % (exchange,))
            self.tile_exchange = exchange

    # ------------------------------------------------------------------------------------------ public API
    def predict_3D(self, x: np.ndarray, do_mirroring: bool, mirror_axes: Tuple[int, ...] = (0, 1, 2),
                   use_sliding_window: bool = False, step_size: float = 0.5, patch_size: Tuple[int, ...] = None,
                   regions_class_order: Tuple[int, ...] = None, use_gaussian: bool = False,
                   pad_border_mode: str = "constant", pad_kwargs: dict = None, all_in_gpu: bool = False,
                   verbose: bool = True, mixed_precision: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        assert step_size <= 1, 'step_size must be smaller than 1. Otherwise there will be a gap between consecutive predictions'
        if verbose:
            print("debug: mirroring", do_mirroring, "mirror_axes", mirror_axes)
        if pad_kwargs is None:
            pad_kwargs = {'constant_values': 0}
        if len(mirror_axes) and max(mirror_axes) > 2:
            raise ValueError("mirror axes. duh")
        if self.training:
            print('WARNING! Network is in train mode during inference. This may be intended, or not...')
        assert len(x.shape) == 4, "data must have shape (c,x,y,z)"
        if self.conv_op != nn.Conv3d:
            raise RuntimeError("Invalid conv op, the MI355X engine implements 3D networks only")
        with torch.no_grad():
            if use_sliding_window:
                return self._internal_predict_3D_3Dconv_tiled(x, step_size, do_mirroring, mirror_axes, patch_size,
                                                              regions_class_order, use_gaussian, pad_border_mode,
                                                              pad_kwargs, all_in_gpu, verbose)
            return self._internal_predict_3D_3Dconv(x, patch_size, do_mirroring, mirror_axes, regions_class_order,
                                                    pad_border_mode, pad_kwargs, verbose)

    @staticmethod
    def _get_gaussian(patch_size, sigma_scale=1. / 8) -> np.ndarray:
        """reference :244-258"""
        tmp = np.zeros(patch_size)
        tmp[tuple(i // 2 for i in patch_size)] = 1
        g = gaussian_filter(tmp, [i * sigma_scale for i in patch_size], 0, mode='constant', cval=0)
        g = (g / np.max(g) * 1).astype(np.float32)
        g[g == 0] = np.min(g[g != 0])
        return g

    @staticmethod
    def _compute_steps_for_sliding_window(patch_size: Tuple[int, ...], image_size: Tuple[int, ...],
                                          step_size: float) -> List[List[int]]:
        """reference :260-284"""
        assert [i >= j for i, j in zip(image_size, patch_size)], "image size must be as large or larger than patch_size"
        assert 0 < step_size <= 1, 'step_size must be larger than 0 and smaller or equal to 1'
        target = [i * step_size for i in patch_size]
        num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, patch_size)]
        steps = []
        for dim in range(len(patch_size)):
            max_step_value = image_size[dim] - patch_size[dim]
            actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
            steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
        return steps

    # ------------------------------------------------------------------------------------------ device helpers
    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("inference on the MI355X engine needs the network on a GPU (no CPU fallback)")
        return dev

    def _inference_nonlin_code(self) -> int:
        """Which fused kernel mode stands for ``self.inference_apply_nonlin`` (the reference applies whatever the attribute
        holds to the network output of every mirrored pass, neural_network.py:531-560; the constructor default is the
        identity, :80, the trainer installs ``softmax_helper``, nnUNetTrainer_simple.py:363): 0 identity, 1 softmax over
        the class axis, 2 sigmoid.  The callable is recognised by what it computes on a probe tensor, so a caller's own
        ``lambda x: F.softmax(x, 1)`` is served as well; anything else raises -- nothing is silently replaced by softmax."""
        fn = self.inference_apply_nonlin
        cached = getattr(self, "_nonlin_probe", None)
        if cached is not None and cached[0] is fn:
            return cached[1]
        from ..utilities.nd_softmax import softmax_helper
        code = None
        if fn is softmax_helper:
            code = 1
        else:
            k = max(2, int(self.num_classes or 2))
            probe = torch.linspace(-3.0, 4.0, k * 8, dtype=torch.float32).reshape(1, k, 2, 2, 2).flip(1) * \
                torch.tensor([1.0, -0.5, 0.25, 2.0, -1.5, 0.75, 1.25, -2.0]).reshape(1, 1, 2, 2, 2)
            try:
                out = fn(probe.clone())
            except Exception as e:      # noqa: BLE001 -- reported with the callable's name below
                out = e
            if isinstance(out, torch.Tensor) and out.shape == probe.shape:
                if torch.equal(out, probe):
                    code = 0
                elif torch.allclose(out, torch.softmax(probe, 1), rtol=1e-6, atol=1e-7):
                    code = 1
                elif torch.allclose(out, torch.sigmoid(probe), rtol=1e-6, atol=1e-7):
                    code = 2
        if code is None:
            raise NotImplementedError(
                "inference_apply_nonlin = %r: the MI355X engine fuses the identity, softmax over the class axis "
                "(softmax_helper) and sigmoid into its mirror-accumulation kernel (e2e_nonlin_flip_acc); apply any other "
                "function to the returned volume instead" % (fn,))
        self._nonlin_probe = (fn, code)
        return code

    def _flip_acc(self, logits: torch.Tensor, result: torch.Tensor, weight: float, first: bool, axes_bits: int):
        """result (+)= weight * flip(inference_apply_nonlin(logits)); HIP kernel (neural_network.py:531-560)."""
        k = logits.shape[-4]
        X, Y, Z = logits.shape[-3:]
        lib().nonlin_flip_acc(logits.data_ptr(), result.data_ptr(), float(weight), 1 if first else 0, k, X, Y, Z,
                              axes_bits, self._inference_nonlin_code(), _stream())

    def _net_probs_into(self, x: torch.Tensor, result: torch.Tensor, weight: float, first: bool, axes_bits: int):
        """result (+)= weight * flip(nonlin(net(x))) with x already flipped."""
        logits = self(x)
        if isinstance(logits, (list, tuple)):
            logits = logits[0]
        self._flip_acc(logits, result, weight, first, axes_bits)

    def _internal_maybe_mirror_and_pred_3D(self, x: Union[np.ndarray, torch.Tensor], mirror_axes: tuple,
                                           do_mirroring: bool = True, mult=None) -> torch.Tensor:
        """reference :500-565.  Returns the [1,K,X,Y,Z] device tensor of (optionally Gaussian-weighted)
        mirrored-and-averaged softmax probabilities."""
        dev = self._device()
        self._inference_nonlin_code()        # (an unsupported callable raises before any forward pass is spent)
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.ascontiguousarray(x)).float()
        x = x.to(dev, non_blocking=True).contiguous()
        assert x.dim() == 5 and x.shape[0] == 1, 'x must be (1, c, x, y, z)'
        c = x.shape[1]
        X, Y, Z = x.shape[2:]
        result = torch.empty((1, self.num_classes, X, Y, Z), dtype=torch.float32, device=dev)
        num_results = 2 ** len(mirror_axes) if do_mirroring else 1
        w = 1 / num_results
        # (flip dims in the reference's order, :529-560) as axis bit sets: bit0 = x (dim 2), bit1 = y, bit2 = z
        combos = [0, 4, 2, 6, 1, 5, 3, 7] if do_mirroring else [0]
        combos = [b for b in combos if all(a in mirror_axes for a in range(3) if b & (1 << a))]
        first = True
        ds = getattr(self, "do_ds", False)
        if hasattr(self, "do_ds"):
            self.do_ds = False
        try:
            if getattr(self, "tta_batched", True) and len(combos) > 1:
                # all mirrored copies in ONE forward pass (batch = number of mirrors): every sample of a batch is computed
                # independently (per-sample tiles, per-sample InstanceNorm), so the logits equal the one-at-a-time ones bit
                # for bit, while the deep, small layers finally get enough workgroups; accumulation order is unchanged
                xb = torch.empty((len(combos),) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
                for m, bits in enumerate(combos):
                    if bits:
                        lib().flip3d(x.data_ptr(), xb[m].data_ptr(), c, X, Y, Z, bits, _stream())
                    else:
                        xb[m].copy_(x[0])
                logits = self(xb)
                if isinstance(logits, (list, tuple)):
                    logits = logits[0]
                for m, bits in enumerate(combos):
                    self._flip_acc(logits[m], result, w, first, bits)
                    first = False
            else:
                flipped = torch.empty_like(x)
                for bits in combos:
                    if bits:
                        lib().flip3d(x.data_ptr(), flipped.data_ptr(), c, X, Y, Z, bits, _stream())
                        self._net_probs_into(flipped, result, w, first, bits)
                    else:
                        self._net_probs_into(x, result, w, first, 0)
                    first = False
        finally:
            if hasattr(self, "do_ds"):
                self.do_ds = ds
        if mult is not None:
            if not isinstance(mult, torch.Tensor):
                mult = torch.from_numpy(mult)
            result *= mult.to(dev)          # host-visible variant; the tiled path fuses this into sw_accumulate
        return result

    # ------------------------------------------------------------------------------------------ tiled prediction
    def _internal_predict_3D_3Dconv_tiled(self, x: np.ndarray, step_size: float, do_mirroring: bool, mirror_axes: tuple,
                                          patch_size: tuple, regions_class_order: tuple, use_gaussian: bool,
                                          pad_border_mode: str, pad_kwargs: dict, all_in_gpu: bool,
                                          verbose: bool) -> Tuple[np.ndarray, np.ndarray]:
        """reference :286-426 (fp32 aggregation like its all_in_gpu=False branch, but resident in HBM)."""
        assert len(x.shape) == 4, "x must be (c, x, y, z)"
        assert patch_size is not None, "patch_size cannot be None for tiled prediction"
        dev = self._device()
        patch_size = tuple(int(p) for p in patch_size)
        data, slicer = pad_nd_image(x, patch_size, pad_border_mode, pad_kwargs, True, None)
        data_shape = data.shape
        steps = self._compute_steps_for_sliding_window(patch_size, data_shape[1:], step_size)
        tiles = [(sx, sy, sz) for sx in steps[0] for sy in steps[1] for sz in steps[2]]
        num_tiles = len(tiles)
        if verbose:
            print("data shape:", data_shape, "patch size:", patch_size, "steps:", steps, "tiles:", num_tiles)

        gauss_dev = None
        if use_gaussian and num_tiles > 1:
            if self._gaussian_3d is None or tuple(self._patch_size_for_gaussian_3d) != patch_size:
                self._gaussian_3d = self._get_gaussian(patch_size, sigma_scale=1. / 8)
                self._patch_size_for_gaussian_3d = patch_size
                self._gaussian_3d_dev = None
            if self._gaussian_3d_dev is None or self._gaussian_3d_dev.device != dev:
                self._gaussian_3d_dev = torch.from_numpy(self._gaussian_3d).to(dev)
            gauss_dev = self._gaussian_3d_dev

        K = self.num_classes
        X, Y, Z = (int(v) for v in data_shape[1:])
        px, py, pz = patch_size
        vol = torch.from_numpy(np.ascontiguousarray(data)).float().to(dev)
        agg = torch.zeros((K, X, Y, Z), dtype=torch.float32, device=dev)
        cnt = torch.zeros((K, X, Y, Z), dtype=torch.float32, device=dev)
        L = lib()

        world, rank = self.tile_world, self.tile_rank

        def predict_tile(ti):
            sx, sy, sz = tiles[ti]
            tile = vol[None, :, sx:sx + px, sy:sy + py, sz:sz + pz].contiguous()
            return self._internal_maybe_mirror_and_pred_3D(tile, mirror_axes, do_mirroring, None)[0]

        def accumulate(ti, patch):
            sx, sy, sz = tiles[ti]
            L.sw_accumulate(patch.data_ptr(), gauss_dev.data_ptr() if gauss_dev is not None else None,
                            agg.data_ptr(), cnt.data_ptr(), K, X, Y, Z, px, py, pz, sx, sy, sz, _stream())

        if world == 1 and not self.tile_force:
            for ti in range(num_tiles):
                accumulate(ti, predict_tile(ti))
        else:
            from ..parallel import run_tiles_sharded, run_tiles_partial
            self.last_shard_stats = {"time": bool(getattr(self, "time_sharding", False))}      # (timing costs a device sync: benchmark only)
            if self.tile_exchange == "allreduce":
                def count_only(ti):
                    sx, sy, sz = tiles[ti]
                    L.sw_accumulate(None, gauss_dev.data_ptr() if gauss_dev is not None else None, agg.data_ptr(), cnt.data_ptr(),
                                    K, X, Y, Z, px, py, pz, sx, sy, sz, _stream())
                self.last_shard_stats["force"] = self.tile_force
                run_tiles_partial(num_tiles, rank, world, self.tile_group, predict_tile, accumulate, count_only, agg,
                                  stats=self.last_shard_stats)
            else:
                run_tiles_sharded(num_tiles, rank, world, self.tile_group, predict_tile, accumulate, (K, px, py, pz), dev,
                                  pipelined=os.environ.get("E2E_SW_BLOCKING") != "1", stats=self.last_shard_stats)

        crop = [(s.start, s.stop) for s in slicer[1:]]
        (cx0, cx1), (cy0, cy1), (cz0, cz1) = crop
        CX, CY, CZ = cx1 - cx0, cy1 - cy0, cz1 - cz0
        probs = torch.empty((K, CX, CY, CZ), dtype=torch.float32, device=dev)
        seg = torch.empty((CX, CY, CZ), dtype=torch.int64, device=dev)
        L.sw_finalize_argmax(agg.data_ptr(), cnt.data_ptr(), probs.data_ptr(), seg.data_ptr(), K, X, Y, Z, cx0, cy0, cz0,
                             CX, CY, CZ, _stream())
        if self.keep_on_device:
            return seg, probs
        probs_np = probs.cpu().numpy()
        if regions_class_order is None:
            seg_np = seg.cpu().numpy()
        else:
            seg_np = np.zeros(probs_np.shape[1:], dtype=np.float32)
            for i, c in enumerate(regions_class_order):
                seg_np[probs_np[i] > 0.5] = c
        if verbose:
            print("prediction done")
        return seg_np, probs_np

    def _internal_predict_3D_3Dconv(self, x: np.ndarray, min_size: Tuple[int, ...], do_mirroring: bool,
                                    mirror_axes: tuple = (0, 1, 2), regions_class_order: tuple = None,
                                    pad_border_mode: str = "constant", pad_kwargs: dict = None,
                                    verbose: bool = True) -> Tuple[np.ndarray, np.ndarray]:
        """reference :464-498: fully convolutional inference (no sliding window)."""
        assert len(x.shape) == 4, "x must be (c, x, y, z)"
        assert self.input_shape_must_be_divisible_by is not None
        data, slicer = pad_nd_image(x, min_size, pad_border_mode, pad_kwargs, True, self.input_shape_must_be_divisible_by)
        pred = self._internal_maybe_mirror_and_pred_3D(data[None], mirror_axes, do_mirroring, None)[0]
        K = pred.shape[0]
        X, Y, Z = pred.shape[1:]
        (cx0, cx1), (cy0, cy1), (cz0, cz1) = [(s.start, s.stop) for s in slicer[1:]]
        ones = torch.ones_like(pred)
        probs = torch.empty((K, cx1 - cx0, cy1 - cy0, cz1 - cz0), dtype=torch.float32, device=pred.device)
        seg = torch.empty(probs.shape[1:], dtype=torch.int64, device=pred.device)
        lib().sw_finalize_argmax(pred.data_ptr(), ones.data_ptr(), probs.data_ptr(), seg.data_ptr(), K, X, Y, Z, cx0, cy0,
                                 cz0, cx1 - cx0, cy1 - cy0, cz1 - cz0, _stream())
        if self.keep_on_device:
            return seg, probs
        probs_np = probs.cpu().numpy()
        if regions_class_order is None:
            return seg.cpu().numpy(), probs_np
        seg_np = np.zeros(probs_np.shape[1:], dtype=np.float32)
        for i, c in enumerate(regions_class_order):
            seg_np[probs_np[i] > 0.5] = c
        return seg_np, probs_np
