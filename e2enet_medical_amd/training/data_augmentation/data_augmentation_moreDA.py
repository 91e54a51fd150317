"""Training input feed on the device.

Mirrors reference e2enet/training/data_augmentation/data_augmentation_moreDA.py:41-209 (``get_moreDA_augmentation``): the
transform list the reference composes from batchgenerators 0.24 (third party, absent from the image) and runs in 24 CPU worker
processes -- SpatialTransform (rotation, scaling, centre crop), GaussianNoise, GaussianBlur, BrightnessMultiplicative,
ContrastAugmentation, SimulateLowResolution, two GammaTransforms, Mirror, Mask, RemoveLabel, DownsampleSegForDSTransform2 --
executed on the GPU by the kernels of csrc/augment.hip.  A raw loader batch ({'data': [B,C,...], 'seg': [B,1,...]} at the
loader's enlarged patch size) goes in, {'data': float32 GPU tensor at the network patch size, 'target': list of
deep-supervision targets} comes out: the host ships the raw patch once and draws ~40 scalars per sample.

Parameter draws follow each transform's own order of random calls as far as it is documented in the cited sources, from ONE
``numpy.random.RandomState(seed)`` per augmenter (the reference's workers each own an unseeded numpy / python RNG: nothing there
to be bit-compatible with).  Parity is unpinned by construction; each kernel is tested against scipy / numpy given the drawn
parameters (tests/test_gpu_augment.py, oracle/augment.py).

Data interpolation follows the reference's orders: ``order_data=3`` (cubic B-spline, scipy ``map_coordinates`` semantics: mirror
prefilter + 64-tap gather) in the spatial transform and ``order_upsample=3`` in SimulateLowResolution (scipy ``zoom`` with
``grid_mode``, edge pre-padding by 12, clipped to the low-resolution range like skimage's ``resize``); ``order_data=1`` selects
the linear kernels.  ``dummy_2D`` (reference :58-60, :80-81; switched on by the plans of anisotropic patches such as BTCV's
48 x 192 x 192, nnUNetTrainer_simple.py:701-716): Convert3DTo2DTransform is a reshape of the batch to [B, C * D, H, W], so the
spatial transform runs on that view as a ONE-slice-deep volume -- one in-plane rotation (``rotation_x``) and scale per sample,
shared by all slices, 2-D cubic B-spline per slice (prefilter along H and W only) -- through the same kernels; the
low-resolution simulation then leaves the slice axis alone (``ignore_axes=(0,)``).  Not built, and rejected when asked for:
elastic deformation (switched off by the trainer, nnUNetTrainer_simple.py:730), the cascade / pyramid transforms, region
targets, additive brightness, independent per-axis scaling, random crops.
"""
from typing import Iterable, Optional, Sequence

import numpy as np
import torch

from ..._lib import lib
from .default_data_augmentation import default_3D_augmentation_params, rotation_matrix_3d
from .downsampling import downsample_seg_for_ds_transform2

OP_NOISE, OP_MUL, OP_CONTRAST, OP_GAMMA_POW, OP_RENORM = 1, 2, 3, 4, 5


def _stream():
    return torch.cuda.current_stream().cuda_stream


def gaussian_weights(sigma: float, truncate: float = 4.0):
    """scipy.ndimage.gaussian_filter1d's kernel: radius = int(truncate * sigma + 0.5), exp(-0.5 x^2 / sigma^2) normalised"""
    r = int(truncate * float(sigma) + 0.5)
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    w /= w.sum()
    return r, w[r:]


class DeviceAugmenter:
    """The train-time transform chain of get_moreDA_augmentation on one device batch."""

    def __init__(self, patch_size: Sequence[int], params=None, border_val_seg=-1, order_seg=1, order_data=3,
                 deep_supervision_scales=None, seed: Optional[int] = None, device="cuda"):
        self.params = dict(default_3D_augmentation_params if params is None else params)
        p = self.params
        if params is None:
            p["do_elastic"] = False        # the table default is True; the trainer, the only caller in the reference, switches it off (:730)
        if p.get("do_elastic"):
            raise NotImplementedError("elastic deformation is not built (the trainer switches it off, nnUNetTrainer_simple.py:730)")
        if p.get("move_last_seg_chanel_to_data"):
            raise NotImplementedError("cascade augmentations are outside the 3D shiftConvPP path")
        self.dummy_2D = bool(p.get("dummy_2D"))
        if p.get("border_mode_data", "constant") != "constant":
            raise NotImplementedError("border_mode_data must be 'constant'")
        if order_seg not in (0, 1):
            raise NotImplementedError("order_seg 0 or 1")
        if order_data not in (1, 3):
            raise NotImplementedError("order_data 1 (linear) or 3 (cubic B-spline, the reference's default)")
        for key, off in (("do_additive_brightness", False), ("independent_scale_factor_for_each_axis", False), ("random_crop", False)):
            if p.get(key, off) not in (off, None):
                raise NotImplementedError("augmentation parameter %s=%r is not built" % (key, p.get(key)))
        self.order_data, self.order_seg = int(order_data), order_seg
        self.order_upsample = 3 if order_data == 3 else 1         # the reference passes order_upsample=3 (:99)
        self.patch_size = tuple(int(v) for v in patch_size)
        self.border_val_seg = float(border_val_seg)
        self.ds_scales = deep_supervision_scales
        self.rs = np.random.RandomState(seed)
        self.device = torch.device(device)
        self.last_draws = None            # the parameters of the last batch (tests, logging)

    # ---- parameter draws (host) --------------------------------------------------------------------------------------------
    def _draw(self, B, C, in_shape):
        p, rs = self.params, self.rs
        d = {"mat": np.zeros((B, 12)), "noise": np.zeros((B, C)), "blur": np.zeros((B, C)), "mul": np.zeros((B, C)),
             "contrast": np.zeros((B, C)), "zoom": np.zeros((B, C)), "gamma_inv": np.zeros((B, C)), "gamma": np.zeros((B, C)),
             "mirror": np.zeros((B, 3), dtype=bool), "modified": np.zeros(B, dtype=bool)}
        shp = np.array(in_shape, dtype=float)
        out_patch = np.array(self.patch_size)
        if self.dummy_2D:                                     # the transform sees [C * D, H, W] images: depth 1 on the device
            shp = np.array([1.0, in_shape[1], in_shape[2]])
            in_shape = (1, in_shape[1], in_shape[2])
            out_patch = np.array([1, self.patch_size[1], self.patch_size[2]])
        for b in range(B):
            # SpatialTransform / augment_spatial: rotation, scaling, then the centre of the loaded patch
            a_mat, modified = np.eye(3), False
            if p.get("do_rotation") and rs.uniform() < p.get("p_rot", 1):
                if self.dummy_2D:                             # dim == 2: one angle (angle_x), rotate_coords_2d
                    a = rs.uniform(*p["rotation_x"]) if rs.uniform() <= p.get("rotation_p_per_axis", 1) else 0.0
                    rot = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
                    a_mat = np.eye(3)
                    a_mat[1:, 1:] = rot.T                     # coords^T R  ==  R^T coords
                else:
                    ang = []
                    for key in ("rotation_x", "rotation_y", "rotation_z"):
                        ang.append(rs.uniform(*p[key]) if rs.uniform() <= p.get("rotation_p_per_axis", 1) else 0.0)
                    a_mat = rotation_matrix_3d(*ang).T        # coords^T R  ==  R^T coords
                modified = True
            if p.get("do_scaling") and rs.uniform() < p.get("p_scale", 1):
                lo, hi = p["scale_range"]
                sc = rs.uniform(lo, 1) if (rs.random_sample() < 0.5 and lo < 1) else rs.uniform(max(lo, 1), hi)
                a_mat = a_mat * sc
                modified = True
            if modified:
                t = shp / 2. - 0.5
            else:                                             # center_crop_aug: integer offsets, an exact copy
                t = (out_patch - 1) / 2. + (np.array(in_shape) - out_patch) // 2
            if self.dummy_2D:                                 # the slice axis is not a spatial axis of the transform
                a_mat[0, :] = 0.0
                a_mat[:, 0] = 0.0
                t[0] = 0.0
            d["mat"][b] = np.concatenate([a_mat, t[:, None]], 1).reshape(-1)
            d["modified"][b] = modified
        for b in range(B):                                    # GaussianNoiseTransform(p_per_sample=0.1), noise_variance (0, 0.1)
            if rs.uniform() < 0.1:
                d["noise"][b, :] = rs.uniform(0, 0.1)
        for b in range(B):                                    # GaussianBlurTransform((0.5, 1), per channel, p 0.2 / 0.5)
            if rs.uniform() < 0.2:
                for c in range(C):
                    if rs.uniform() <= 0.5:
                        d["blur"][b, c] = rs.uniform(0.5, 1.0)
        for b in range(B):                                    # BrightnessMultiplicativeTransform((0.75, 1.25), p 0.15)
            if rs.uniform() < 0.15:
                d["mul"][b] = rs.uniform(0.75, 1.25, size=C)
        for b in range(B):                                    # ContrastAugmentationTransform((0.75, 1.25), p 0.15)
            if rs.uniform() < 0.15:
                for c in range(C):
                    d["contrast"][b, c] = rs.uniform(0.75, 1) if rs.random_sample() < 0.5 else rs.uniform(1, 1.25)
        for b in range(B):                                    # SimulateLowResolutionTransform((0.5, 1), p 0.25 / 0.5)
            if rs.uniform() < 0.25:
                for c in range(C):
                    if rs.uniform() < 0.5:
                        d["zoom"][b, c] = rs.uniform(0.5, 1)
        glo, ghi = p.get("gamma_range", (0.7, 1.5))

        def gamma_draw():
            return rs.uniform(glo, 1) if (rs.random_sample() < 0.5 and glo < 1) else rs.uniform(max(glo, 1), ghi)
        for b in range(B):                                    # GammaTransform(invert_image=True, p 0.1)
            if rs.uniform() < 0.1:
                d["gamma_inv"][b] = [gamma_draw() for _ in range(C)]
        if p.get("do_gamma"):
            for b in range(B):                                # GammaTransform(invert_image=False, p_gamma)
                if rs.uniform() < p["p_gamma"]:
                    d["gamma"][b] = [gamma_draw() for _ in range(C)]
        if p.get("do_mirror") or p.get("mirror"):
            for b in range(B):                                # MirrorTransform: every axis with probability 0.5
                for ax in (0, 1, 2):
                    if ax in p.get("mirror_axes", (0, 1, 2)) and rs.uniform() < 0.5:
                        d["mirror"][b, ax] = True
        return d

    # ---- device execution ------------------------------------------------------------------------------------------------------
    def _prm(self, rows):
        return torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float64)).to(self.device)

    def _stats(self, x, nbc, vol):
        L = lib()
        out = torch.empty(nbc * 4, dtype=torch.float64, device=self.device)
        ws = torch.empty(L.aug_stats_ws_bytes(nbc) // 8, dtype=torch.float64, device=self.device)
        L.aug_stats(x.data_ptr(), out.data_ptr(), ws.data_ptr(), nbc, vol, _stream())
        return out.cpu().numpy().reshape(nbc, 4)

    def apply(self, data: torch.Tensor, seg: Optional[torch.Tensor], draws: dict):
        """Run the chain with the given parameters.  data [B,C,D,H,W], seg [B,CS,D,H,W] float32 on the device."""
        L = lib()
        B0, C0, D0 = int(data.shape[0]), int(data.shape[1]), int(data.shape[2])
        if self.dummy_2D:                                     # Convert3DTo2DTransform: [B, C, D, H, W] -> [B, C * D, (1,) H, W]
            assert D0 == self.patch_size[0], "dummy_2D: the loader's patch keeps the network's depth (basic_generator_patch_size)"
            data = data.reshape(B0, C0 * D0, 1, data.shape[3], data.shape[4])
            if seg is not None:
                seg = seg.reshape(B0, seg.shape[1] * D0, 1, seg.shape[3], seg.shape[4])
        B, C = data.shape[:2]
        Di, Hi, Wi = (int(v) for v in data.shape[2:])
        Do, Ho, Wo = (1, self.patch_size[1], self.patch_size[2]) if self.dummy_2D else self.patch_size
        vol = Do * Ho * Wo
        nbc = B * C
        out = torch.empty((B, C, Do, Ho, Wo), dtype=torch.float32, device=self.device)
        oseg = torch.empty((B, seg.shape[1], Do, Ho, Wo), dtype=torch.float32, device=self.device) if seg is not None else None
        mat = self._prm(draws["mat"])
        coef = cubic = None
        mod = np.asarray(draws["modified"], dtype=bool)
        if self.order_data == 3 and mod.any():
            # cubic B-spline coefficients of the samples the transform moves (an untouched sample is an integer crop: exact copy).
            # scipy's prefilter, one recursive pass per axis; the raw batch stays as it is (order-1 seg path, centre crops)
            coef = torch.empty_like(data)
            for b in np.nonzero(mod)[0]:
                src = data[b]
                for ax in ((1, 2) if self.dummy_2D else (0, 1, 2)):       # (a one-slice axis has nothing to filter)
                    L.aug_bspline_prefilter_axis(src.data_ptr(), coef[b].data_ptr(), C, Di, Hi, Wi, ax, _stream())
                    src = coef[b]
            cubic = torch.from_numpy(mod.astype(np.int32)).to(self.device)
        L.aug_spatial(data.data_ptr(), coef.data_ptr() if coef is not None else None, cubic.data_ptr() if cubic is not None else None,
                      seg.data_ptr() if seg is not None else None, out.data_ptr(),
                      oseg.data_ptr() if oseg is not None else None, mat.data_ptr(), B, C, seg.shape[1] if seg is not None else 0,
                      Di, Hi, Wi, Do, Ho, Wo, self.order_seg, self.border_val_seg, _stream())
        # a sample the spatial transform did not touch is a centre crop: exact for data and seg alike (order-1 seg at integer
        # coordinates is the voxel's own label, except that batchgenerators' crop keeps -1 where the order-1 path writes 0)
        if seg is not None and self.order_seg == 1 and not draws["modified"].all():
            for b in np.nonzero(~draws["modified"])[0]:
                lo = [(i - o) // 2 for i, o in zip((Di, Hi, Wi), (Do, Ho, Wo))]
                oseg[b].copy_(seg[b, :, lo[0]:lo[0] + Do, lo[1]:lo[1] + Ho, lo[2]:lo[2] + Wo])
        if self.dummy_2D:                                     # Convert2DTo3DTransform: back to [B, C, D, H, W]; the rest of the chain is 3-D
            out = out.reshape(B0, C0, D0, Ho, Wo)
            if oseg is not None:
                oseg = oseg.reshape(B0, oseg.shape[1] // D0, D0, Ho, Wo)
            B, C, Do = B0, C0, D0
            vol = Do * Ho * Wo
            nbc = B * C
        tmp = None

        def rows(active, cols):
            r = np.zeros((nbc, 8))
            r[:, 0] = active.reshape(-1)
            for k, v in cols.items():
                r[:, k] = np.asarray(v, dtype=np.float64).reshape(-1)
            return r
        noise = draws["noise"]
        if (noise > 0).any():
            self._noise_seed = int(self.rs.randint(0, 2 ** 31 - 1)) if "noise_seed" not in draws else int(draws["noise_seed"])
            L.aug_pointwise(out.data_ptr(), self._prm(rows(noise > 0, {1: noise})).data_ptr(), OP_NOISE, nbc, vol,
                            self._noise_seed, _stream())
        blur = draws["blur"]
        if (blur > 0).any():
            w = np.zeros((nbc, 16), dtype=np.float32)
            for i, s in enumerate(blur.reshape(-1)):
                if s > 0:
                    r, ww = gaussian_weights(s)
                    w[i, 0] = r
                    w[i, 1:2 + r] = ww
            wt = torch.from_numpy(w).to(self.device)
            tmp = torch.empty_like(out)
            src, dst = out, tmp
            for ax in range(3):
                L.aug_blur_axis(src.data_ptr(), dst.data_ptr(), wt.data_ptr(), nbc, Do, Ho, Wo, ax, _stream())
                src, dst = dst, src
            out, tmp = src, dst
        mul = draws["mul"]
        if (mul != 0).any():
            L.aug_pointwise(out.data_ptr(), self._prm(rows(mul != 0, {1: mul})).data_ptr(), OP_MUL, nbc, vol, 0, _stream())
        con = draws["contrast"]
        if (con != 0).any():
            st = self._stats(out, nbc, vol)
            L.aug_pointwise(out.data_ptr(), self._prm(rows(con != 0, {1: con, 2: st[:, 2], 3: st[:, 0], 4: st[:, 1]})).data_ptr(),
                            OP_CONTRAST, nbc, vol, 0, _stream())
        zoom = draws["zoom"]
        if (zoom > 0).any():
            lo = np.zeros((nbc, 3), dtype=np.int32)
            for i, z in enumerate(zoom.reshape(-1)):
                if z > 0:
                    lo[i] = np.round(np.array([Do, Ho, Wo]) * z).astype(int)
                    if self.dummy_2D:
                        lo[i, 0] = Do                         # ignore_axes=(0,): the slice axis keeps its resolution
            if tmp is None:
                tmp = torch.empty_like(out)
            if self.order_upsample == 3:
                # per touched channel: nearest down-sampling into an edge-padded scratch volume, its range (skimage clips the
                # up-sampled result to it), spline prefilter, cubic up-sampling; untouched channels are copied
                PAD = 12
                tmp.copy_(out)
                ov, tv = out.view(nbc, Do, Ho, Wo), tmp.view(nbc, Do, Ho, Wo)
                big = int(max((l[0] + 2 * PAD) * (l[1] + 2 * PAD) * (l[2] + 2 * PAD) for l in lo))
                scratch = torch.empty(big, dtype=torch.float32, device=self.device)
                st = torch.empty(4, dtype=torch.float64, device=self.device)
                ws = torch.empty(L.aug_stats_ws_bytes(1) // 8, dtype=torch.float64, device=self.device)
                for i in np.nonzero(lo[:, 0])[0]:
                    ld, lh, lw = (int(v) for v in lo[i])
                    pd, ph, pw = ld + 2 * PAD, lh + 2 * PAD, lw + 2 * PAD
                    L.aug_lowres_down(ov[i].data_ptr(), scratch.data_ptr(), Do, Ho, Wo, ld, lh, lw, PAD, _stream())
                    L.aug_stats(scratch.data_ptr(), st.data_ptr(), ws.data_ptr(), 1, pd * ph * pw, _stream())
                    for ax in (0, 1, 2):
                        L.aug_bspline_prefilter_axis(scratch.data_ptr(), scratch.data_ptr(), 1, pd, ph, pw, ax, _stream())
                    L.aug_lowres_up3(scratch.data_ptr(), tv[i].data_ptr(), st.data_ptr(), Do, Ho, Wo, ld, lh, lw, PAD, _stream())
            else:
                L.aug_lowres(out.data_ptr(), tmp.data_ptr(), torch.from_numpy(lo).to(self.device).data_ptr(), nbc, Do, Ho, Wo, _stream())
            out, tmp = tmp, out
        for key, inv in (("gamma_inv", 1.0), ("gamma", 0.0)):
            g = draws[key]
            if not (g != 0).any():
                continue
            act = g != 0
            st = self._stats(out, nbc, vol)                       # statistics of the data (of -data when inverted)
            if inv:
                mn_, mx_, mean_ = -st[:, 1], -st[:, 0], -st[:, 2]
            else:
                mn_, mx_, mean_ = st[:, 0], st[:, 1], st[:, 2]
            sd_ = st[:, 3]
            L.aug_pointwise(out.data_ptr(), self._prm(rows(act, {1: g, 2: mn_, 3: mx_ - mn_, 5: inv})).data_ptr(), OP_GAMMA_POW, nbc,
                            vol, 0, _stream())
            if self.params.get("gamma_retain_stats"):
                st2 = self._stats(out, nbc, vol)
                mean2 = -st2[:, 2] if inv else st2[:, 2]
                L.aug_pointwise(out.data_ptr(), self._prm(rows(act, {1: mean2, 2: st2[:, 3], 3: mean_, 4: sd_, 5: inv})).data_ptr(),
                                OP_RENORM, nbc, vol, 0, _stream())
        mir = draws["mirror"]
        if mir.any():
            if tmp is None:
                tmp = torch.empty_like(out)
            stmp = torch.empty_like(oseg) if oseg is not None else None
            for b in range(B):
                axes = int(mir[b, 0]) | (int(mir[b, 1]) << 1) | (int(mir[b, 2]) << 2)
                if axes == 0:
                    continue
                L.flip3d(out[b].data_ptr(), tmp[b].data_ptr(), C, Do, Ho, Wo, axes, _stream())
                out[b].copy_(tmp[b])
                if oseg is not None:
                    L.flip3d(oseg[b].data_ptr(), stmp[b].data_ptr(), oseg.shape[1], Do, Ho, Wo, axes, _stream())
                    oseg[b].copy_(stmp[b])
        if oseg is not None:
            um = self.params.get("mask_was_used_for_normalization")
            um_t = None
            if um is not None:
                um_t = torch.tensor([1 if um[c] else 0 for c in range(C)], dtype=torch.int32, device=self.device)
            L.aug_finish(out.data_ptr(), oseg.data_ptr(), um_t.data_ptr() if um_t is not None else None, B, C, oseg.shape[1], vol,
                         _stream())
        return out, oseg

    def __call__(self, data, seg=None):
        """data / seg: numpy arrays or tensors [B, C, ...] at the loader's patch size.  Returns {'data', 'target'}."""
        data = torch.as_tensor(data, dtype=torch.float32).to(self.device, non_blocking=True).contiguous()
        if seg is not None:
            seg = torch.as_tensor(seg, dtype=torch.float32).to(self.device, non_blocking=True).contiguous()
            if self.params.get("selected_seg_channels") is not None:
                seg = seg[:, list(self.params["selected_seg_channels"])].contiguous()
        if self.params.get("selected_data_channels") is not None:
            data = data[:, list(self.params["selected_data_channels"])].contiguous()
        draws = self._draw(data.shape[0], data.shape[1], data.shape[2:])
        self.last_draws = draws
        out, oseg = self.apply(data, seg, draws)
        res = {"data": out}
        if oseg is not None:
            res["target"] = (downsample_seg_for_ds_transform2(oseg, self.ds_scales, order=0) if self.ds_scales is not None else oseg)
        return res


class _DeviceGenerator:
    """Wraps a loader (iterable of {'data', 'seg'} numpy batches) -- what MultiThreadedAugmenter(dataloader, Compose(...)) is in
    the reference -- and yields device batches."""

    def __init__(self, loader: Iterable, fn):
        self.loader, self.fn, self._it = loader, fn, None

    def __iter__(self):
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self.loader)
        b = next(self._it)
        # DataLoader3D hands its page-locked staging tensors along: the upload below is then an asynchronous copy; the loader is told
        # when that copy is behind us (an event on this stream) and does not refill the buffer earlier
        res = self.fn(b.get("data_pinned", b["data"]), b.get("seg_pinned", b.get("seg")))
        if "data_pinned" in b and hasattr(self.loader, "mark_uploaded") and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.loader.mark_uploaded(ev)
        return res

    next = __next__


def get_moreDA_augmentation(dataloader_train, dataloader_val, patch_size, params=default_3D_augmentation_params,
                            border_val_seg=-1, seeds_train=None, seeds_val=None, order_seg=1, order_data=3,
                            deep_supervision_scales=None, soft_ds=False, classes=None, pin_memory=True, regions=None,
                            use_nondetMultiThreadedAugmenter: bool = False):
    """Same signature as the reference (:41-46).  Returns (train generator, validation generator); the validation chain is
    RemoveLabel + target down-sampling only (:152-172)."""
    assert params.get('mirror') is None, "old version of params, use new keyword do_mirror"
    if soft_ds or regions is not None:
        raise NotImplementedError("soft deep-supervision targets / region targets are not built")
    seed = None if seeds_train is None else int(np.asarray(seeds_train).reshape(-1)[0])
    aug = DeviceAugmenter(patch_size, params, border_val_seg, order_seg, order_data, deep_supervision_scales, seed)

    def val_fn(data, seg):
        dev = aug.device
        data = torch.as_tensor(data, dtype=torch.float32).to(dev).contiguous()
        seg = torch.as_tensor(seg, dtype=torch.float32).to(dev).contiguous()
        if params.get("selected_data_channels") is not None:              # DataChannelSelectionTransform (:154-155)
            data = data[:, list(params["selected_data_channels"])].contiguous()
        if params.get("selected_seg_channels") is not None:
            seg = seg[:, list(params["selected_seg_channels"])].contiguous()
        lib().aug_finish(data.data_ptr(), seg.data_ptr(), None, data.shape[0], data.shape[1], seg.shape[1],
                         int(np.prod(seg.shape[2:])), _stream())            # RemoveLabelTransform(-1, 0)
        tgt = downsample_seg_for_ds_transform2(seg, deep_supervision_scales, order=0) if deep_supervision_scales is not None else seg
        return {"data": data, "target": tgt}
    return _DeviceGenerator(dataloader_train, aug), _DeviceGenerator(dataloader_val, val_fn)
