"""the training input feed on the device (csrc/augment.hip, data_augmentation_moreDA.DeviceAugmenter).
Parity is unpinned by construction (batchgenerators is absent): every kernel is checked, given the drawn parameters, against the
scipy / numpy restatement in oracle/augment.py."""
import numpy as np
import pytest
import torch

from oracle import augment as oaug

pytestmark = pytest.mark.gpu


def _aug(patch=(24, 32, 28), **kw):
    from e2enet_medical_amd.training.data_augmentation.data_augmentation_moreDA import DeviceAugmenter
    from e2enet_medical_amd.training.data_augmentation.default_data_augmentation import default_3D_augmentation_params
    p = dict(default_3D_augmentation_params)
    p.update(do_elastic=False, scale_range=(0.7, 1.4), selected_seg_channels=[0])
    p.update(kw.pop("params", {}))
    return DeviceAugmenter(patch, p, seed=kw.pop("seed", 0), **kw)


def _raw(B=2, C=3, shape=(40, 48, 44), seed=0, labels=4):
    rng = np.random.RandomState(seed)
    data = rng.standard_normal((B, C) + shape).astype(np.float32)
    # blocky label map with a -1 region (nnU-Net marks voxels outside the nonzero mask with -1)
    seg = rng.randint(0, labels, (B, 1) + tuple(s // 4 for s in shape)).astype(np.float32)
    seg = np.kron(seg, np.ones((1, 1, 4, 4, 4), np.float32))
    seg[:, :, :6] = -1
    return data, seg


def _blank(a, B, C):
    from e2enet_medical_amd.training.data_augmentation.data_augmentation_moreDA import DeviceAugmenter  # noqa: F401
    return {"mat": np.zeros((B, 12)), "noise": np.zeros((B, C)), "blur": np.zeros((B, C)), "mul": np.zeros((B, C)),
            "contrast": np.zeros((B, C)), "zoom": np.zeros((B, C)), "gamma_inv": np.zeros((B, C)), "gamma": np.zeros((B, C)),
            "mirror": np.zeros((B, 3), dtype=bool), "modified": np.ones(B, dtype=bool)}


def _mats(a, B, in_shape, angles, scales):
    from e2enet_medical_amd.training.data_augmentation.default_data_augmentation import rotation_matrix_3d
    m = np.zeros((B, 12))
    for b in range(B):
        A = rotation_matrix_3d(*angles[b]).T * scales[b]
        t = np.array(in_shape, dtype=float) / 2. - 0.5
        m[b] = np.concatenate([A, t[:, None]], 1).reshape(-1)
    return m


@pytest.mark.parametrize("order_seg,order_data", [(0, 3), (1, 3), (1, 1)])
def test_spatial_transform_vs_scipy(order_seg, order_data):
    """order_data 3 = the reference's cubic B-spline (data_augmentation_moreDA.py:43): scipy's mirror prefilter and 64-tap gather
    on the device against scipy.ndimage.map_coordinates itself; order 1 = the linear mode."""
    a = _aug(order_seg=order_seg, order_data=order_data)
    data, seg = _raw()
    d = _blank(a, 2, 3)
    d["mat"] = _mats(a, 2, data.shape[2:], [(0.3, -0.2, 0.4), (0.0, 0.5, -0.1)], [1.3, 0.75])
    out, oseg = a.apply(torch.from_numpy(data).cuda(), torch.from_numpy(seg).cuda(), d)
    ref, rseg = oaug.spatial(data, seg, d["mat"], a.patch_size, order_seg, -1.0, order_data)
    rdat, rseg = oaug.finish(ref, rseg, None)
    err = np.abs(out.cpu().numpy() - rdat).max()
    print("[spatial order_data %d] max |device - scipy| = %.3e" % (order_data, err))
    assert err <= (2e-5 if order_data == 1 else 3e-5)          # (coefficients are stored in fp32 between the prefilter passes)
    assert (out.cpu().numpy() == 0).mean() > 0.01               # part of the patch lies outside the loaded volume: cval
    assert (oseg.cpu().numpy() != rseg).mean() <= 2e-4          # label decisions exactly on a 0.5 / half-voxel boundary


@pytest.mark.parametrize("shape", [(3, 9, 11, 13), (2, 24, 40, 330), (1, 40, 5, 64), (2, 1, 3, 700)])
def test_bspline_prefilter_is_scipy_spline_filter(shape):
    """e2e_aug_bspline_prefilter_axis x 3 = scipy.ndimage.spline_filter(order=3, mode='mirror') (what map_coordinates / zoom run in
    front of an order-3 interpolation, also for mode='constant' and the pre-padded 'nearest'); rows longer than 320 voxels take the
    16-lines-per-block form of the contiguous-axis kernel, a length-1 axis is left alone."""
    from scipy import ndimage
    from e2enet_medical_amd._lib import lib
    x = np.random.RandomState(5).standard_normal(shape).astype(np.float32)
    t = torch.from_numpy(x).cuda()
    o = torch.empty_like(t)
    src = t
    for ax in (0, 1, 2):
        lib().aug_bspline_prefilter_axis(src.data_ptr(), o.data_ptr(), shape[0], *shape[1:], ax, 0)
        src = o
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.from_numpy(x))            # the source is not touched
    for v in range(shape[0]):
        ref = ndimage.spline_filter(x[v].astype(np.float64), 3, mode='mirror')
        assert np.abs(o[v].cpu().numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_unmodified_sample_is_a_centre_crop():
    a = _aug()
    data, seg = _raw()
    d = a._draw(2, 3, data.shape[2:])
    d = dict(_blank(a, 2, 3), mat=d["mat"], modified=np.zeros(2, dtype=bool))
    for b in range(2):          # identity + centre offset, as _draw builds it for an unmodified sample
        lo = (np.array(data.shape[2:]) - np.array(a.patch_size)) // 2
        d["mat"][b] = np.concatenate([np.eye(3), ((np.array(a.patch_size) - 1) / 2. + lo)[:, None]], 1).reshape(-1)
    out, oseg = a.apply(torch.from_numpy(data).cuda(), torch.from_numpy(seg).cuda(), d)
    lo = [(i - o) // 2 for i, o in zip(data.shape[2:], a.patch_size)]
    sl = (slice(None), slice(None)) + tuple(slice(l, l + o) for l, o in zip(lo, a.patch_size))
    assert np.array_equal(out.cpu().numpy(), data[sl])
    want = seg[sl].copy()
    want[want == -1] = 0
    assert np.array_equal(oseg.cpu().numpy(), want)


def test_intensity_transforms_vs_numpy_scipy():
    a = _aug()
    data, seg = _raw(shape=(24, 32, 28))
    B, C = 2, 3
    ident = np.zeros((B, 12))
    for b in range(B):
        ident[b] = np.concatenate([np.eye(3), ((np.array(a.patch_size) - 1) / 2.)[:, None]], 1).reshape(-1)
    x = torch.from_numpy(data).cuda()
    s = torch.from_numpy(seg).cuda()

    def run(**kw):
        d = dict(_blank(a, B, C), mat=ident, modified=np.zeros(B, dtype=bool))      # an untouched sample: exact integer crop
        d.update(kw)
        return a.apply(x, s, d)[0].cpu().numpy()
    # blur (per channel sigma, one channel untouched)
    sig = np.array([[0.6, 0.0, 1.0], [0.0, 0.85, 0.5]])
    got = run(blur=sig)
    for b in range(B):
        for c in range(C):
            ref = data[b, c] if sig[b, c] == 0 else oaug.gaussian_blur(data[b, c], sig[b, c])
            assert np.abs(got[b, c] - ref).max() <= 2e-6
    # multiplicative brightness
    mul = np.array([[0.8, 1.2, 0.0], [1.1, 0.0, 0.9]])
    got = run(mul=mul)
    for b in range(B):
        for c in range(C):
            ref = data[b, c] if mul[b, c] == 0 else (data[b, c].astype(np.float64) * mul[b, c]).astype(np.float32)
            assert np.array_equal(got[b, c], ref)
    # contrast
    con = np.array([[0.8, 0.0, 1.2], [0.0, 1.1, 0.9]])
    got = run(contrast=con)
    for b in range(B):
        for c in range(C):
            ref = data[b, c] if con[b, c] == 0 else oaug.contrast(data[b, c], con[b, c])
            assert np.abs(got[b, c] - ref).max() <= 2e-6
    # gamma, plain and inverted, with retain_stats
    g = np.array([[0.8, 1.3, 0.0], [0.0, 0.75, 1.45]])
    for key, inv in (("gamma", False), ("gamma_inv", True)):
        got = run(**{key: g})
        for b in range(B):
            for c in range(C):
                ref = data[b, c] if g[b, c] == 0 else oaug.gamma(data[b, c], g[b, c], inv, True)
                assert np.abs(got[b, c] - ref).max() <= 5e-5, (key, b, c, np.abs(got[b, c] - ref).max())
    # low-resolution simulation: the reference's cubic up-sampling (order_upsample = 3, clipped to the low-resolution range), and linear
    z = np.array([[0.5, 0.0, 0.8], [0.66, 0.93, 0.0]])
    for order, aa in ((3, a), (1, _aug(order_data=1))):
        d = dict(_blank(aa, B, C), mat=ident, zoom=z, modified=np.zeros(B, dtype=bool))
        got = aa.apply(x, s, d)[0].cpu().numpy()
        for b in range(B):
            for c in range(C):
                ref = data[b, c] if z[b, c] == 0 else oaug.low_resolution(data[b, c], z[b, c], order)
                assert np.abs(got[b, c] - ref).max() <= (2e-6 if order == 1 else 2e-5), (order, b, c, np.abs(got[b, c] - ref).max())
        if order == 3:                                          # the clip matters here: the cubic overshoots the low-resolution range somewhere
            lo = oaug._resize(data[0, 0], np.round(np.array(data.shape[2:]) * 0.5).astype(int), 0)
            raw = oaug._resize(lo, data.shape[2:], 3)
            assert (raw > lo.max()).any() or (raw < lo.min()).any()
            assert got[0, 0].max() <= np.float32(lo.max()) and got[0, 0].min() >= np.float32(lo.min())
    # mirror + mask + remove label
    a2 = _aug(params={"mask_was_used_for_normalization": {0: True, 1: False, 2: True}})
    d = dict(_blank(a2, B, C), mat=ident, mirror=np.array([[True, False, True], [False, True, False]]), modified=np.zeros(B, dtype=bool))
    out, oseg = a2.apply(x, s, d)
    for b in range(B):
        rd, rs = oaug.mirror(data[b], seg[b], d["mirror"][b])
        rd, rs = oaug.finish(rd[None], rs[None], [True, False, True])
        assert np.array_equal(out[b].cpu().numpy(), rd[0]) and np.array_equal(oseg[b].cpu().numpy(), rs[0])


def test_gaussian_noise_statistics_and_determinism():
    a = _aug(patch=(32, 32, 32))
    data = np.zeros((1, 2, 32, 32, 32), np.float32)
    ident = np.concatenate([np.eye(3), np.full((3, 1), 15.5)], 1).reshape(1, 12)
    d = dict(_blank(a, 1, 2), mat=ident, noise=np.array([[0.07, 0.0]]), noise_seed=123, modified=np.zeros(1, dtype=bool))
    x = torch.from_numpy(data).cuda()
    o1 = a.apply(x, None, d)[0].cpu().numpy()
    o2 = a.apply(x, None, d)[0].cpu().numpy()
    assert np.array_equal(o1, o2)
    assert np.all(o1[0, 1] == 0)
    n = o1[0, 0].astype(np.float64)
    assert abs(n.mean()) < 3 * 0.07 / np.sqrt(n.size) + 1e-4 and abs(n.std() - 0.07) < 0.07 * 0.02
    k = ((n - n.mean()) ** 4).mean() / n.var() ** 2
    assert abs(k - 3.0) < 0.15                                  # Gaussian kurtosis
    d2 = dict(d, noise_seed=124)
    assert not np.array_equal(o1, a.apply(x, None, d2)[0].cpu().numpy())


def test_full_chain_runs_from_a_seed_and_feeds_the_trainer_shapes():
    """DeviceAugmenter.__call__: raw loader batch -> {'data', 'target' list}; the same seed reproduces the batch."""
    scales = [[1, 1, 1], [0.5, 0.5, 0.5], [0.25, 0.25, 0.25]]
    data, seg = _raw(B=2, C=2, shape=(40, 48, 44), seed=3)
    outs = []
    for _ in range(2):
        a = _aug(patch=(24, 32, 32), deep_supervision_scales=scales, seed=11)
        res = [a(data, seg) for _ in range(6)]                 # several batches: every transform fires at least once with high probability
        outs.append(res)
    for r1, r2 in zip(*outs):
        assert torch.equal(r1["data"], r2["data"]) and all(torch.equal(t1, t2) for t1, t2 in zip(r1["target"], r2["target"]))
    r = outs[0][0]
    assert r["data"].shape == (2, 2, 24, 32, 32) and r["data"].is_cuda and torch.isfinite(r["data"]).all()
    assert [tuple(t.shape) for t in r["target"]] == [(2, 1, 24, 32, 32), (2, 1, 12, 16, 16), (2, 1, 6, 8, 8)]
    assert float(r["target"][0].min()) >= 0.0                  # RemoveLabelTransform(-1, 0)


@pytest.mark.parametrize("order_seg,order_data", [(1, 3), (0, 1)])
def test_dummy_2d_spatial_transform_vs_scipy_per_slice(order_seg, order_data):
    """dummy_2D (reference data_augmentation_moreDA.py:58-60, :80-81; switched on for anisotropic patches such as BTCV's
    48 x 192 x 192 by nnUNetTrainer_simple.py:701-716): the batch is viewed as [B, C * D, H, W], ONE in-plane rotation + scale per
    sample, every slice interpolated in 2-D (cubic B-spline with a prefilter along H and W only), the slice axis untouched.
    Against scipy.ndimage.map_coordinates slice by slice."""
    patch = (12, 32, 28)
    a = _aug(patch=patch, order_seg=order_seg, order_data=order_data, params={"dummy_2D": True, "rotation_x": (-np.pi, np.pi)})
    data, seg = _raw(B=2, C=2, shape=(12, 48, 44), seed=7)
    d = _blank(a, 2, 2)
    mats = np.zeros((2, 12))
    for b, (ang, sc) in enumerate([(0.4, 1.25), (-1.1, 0.8)]):
        A = np.zeros((3, 3))
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        A[1:, 1:] = rot.T * sc
        t = np.array([0.0, data.shape[3] / 2. - 0.5, data.shape[4] / 2. - 0.5])
        mats[b] = np.concatenate([A, t[:, None]], 1).reshape(-1)
    d["mat"] = mats
    out, oseg = a.apply(torch.from_numpy(data).cuda(), torch.from_numpy(seg).cuda(), d)
    assert tuple(out.shape) == (2, 2) + patch and tuple(oseg.shape) == (2, 1) + patch
    ref, rseg = oaug.spatial_dummy_2d(data, seg, mats, patch, order_seg, -1.0, order_data)
    rdat, rseg = oaug.finish(ref, rseg, None)
    err = np.abs(out.cpu().numpy() - rdat).max()
    print("[dummy_2D spatial order_data %d] max |device - scipy| = %.3e" % (order_data, err))
    assert err <= 3e-5
    assert (oseg.cpu().numpy() != rseg).mean() <= 2e-4


def test_dummy_2d_draws_low_resolution_and_full_chain():
    """dummy_2D end to end: the draws rotate in-plane only (slice row / column of the affine zero), SimulateLowResolution keeps the
    slice axis (ignore_axes=(0,): against the scipy restatement), and the whole chain runs on the BTCV-like anisotropic shape the
    reference's planner switches dummy_2D on for (max(patch) / patch[0] > 3)."""
    patch = (12, 48, 48)
    scales = [[1, 1, 1], [1, 0.5, 0.5], [0.5, 0.25, 0.25]]
    a = _aug(patch=patch, deep_supervision_scales=scales, seed=5, params={"dummy_2D": True, "rotation_x": (-np.pi, np.pi)})
    data, seg = _raw(B=2, C=1, shape=(12, 64, 60), seed=9)
    seen_mod = False
    for _ in range(6):
        r = a(data, seg)
        m = a.last_draws["mat"].reshape(2, 3, 4)
        assert np.all(m[:, 0, :] == 0) and np.all(m[:, :, 0] == 0)
        seen_mod |= bool(a.last_draws["modified"].any())
        assert r["data"].shape == (2, 1) + patch and torch.isfinite(r["data"]).all()
        assert [tuple(t.shape) for t in r["target"]] == [(2, 1, 12, 48, 48), (2, 1, 12, 24, 24), (2, 1, 6, 12, 12)]
    assert seen_mod
    # low-resolution simulation alone
    x = np.random.RandomState(3).standard_normal((1, 1) + patch).astype(np.float32)
    d = _blank(a, 1, 1)
    ident = np.zeros((3, 4)); ident[1, 1] = ident[2, 2] = 1.0
    ident[1, 3], ident[2, 3] = (patch[1] - 1) / 2., (patch[2] - 1) / 2.
    d["mat"] = ident.reshape(1, 12)
    d["modified"] = np.zeros(1, dtype=bool)
    d["zoom"] = np.array([[0.6]])
    out, _ = a.apply(torch.from_numpy(x).cuda(), None, d)
    ref = oaug.low_resolution(x[0, 0], 0.6, 3, ignore_axes=(0,))
    assert np.abs(out.cpu().numpy()[0, 0] - ref).max() <= 5e-5
