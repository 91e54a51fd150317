"""Shared test helpers: closed-form weights (so golden fixtures carry outputs only),
seeded inputs, fixture loading."""
import math
import os
import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name), allow_pickle=False)


def closed_form_tensor(shape, index, kind):
    """Deterministic pseudo-weights: value(j) = amp * sin(0.618 j + 1.3 index + 0.5) (+ offset).
    kind: 'conv' (He-scaled), 'gamma' (around 1), 'beta' / 'bias' (small)."""
    n = int(np.prod(shape))
    j = np.arange(n, dtype=np.float64)
    base = np.sin(0.618 * j + 1.3 * index + 0.5)
    if kind == "conv":
        fan_in = shape[1] * int(np.prod(shape[2:]))
        amp = math.sqrt(2.0 / (1 + 0.01 ** 2)) / math.sqrt(fan_in) * math.sqrt(2.0)
        v = amp * base
    elif kind == "gamma":
        v = 1.0 + 0.1 * base
    elif kind == "beta":
        v = 0.1 * base
    else:
        v = 0.05 * base
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def param_kind(name):
    if name.endswith("instnorm.weight"):
        return "gamma"
    if name.endswith("instnorm.bias"):
        return "beta"
    if name.endswith("conv.bias"):
        return "bias"
    return "conv"


def closed_form_params(shapes):
    """shapes: ordered dict name -> shape."""
    return {name: closed_form_tensor(tuple(shp), i, param_kind(name))
            for i, (name, shp) in enumerate(shapes.items())}


def seeded_input(shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32)


def seeded_labels(shape, num_classes, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, num_classes, shape, generator=g).float()


def pack_kernel_mask(mask: torch.Tensor) -> np.ndarray:
    """[out, in, k, k, k] 0/1 mask -> packed bits of the [out, in] kernel map."""
    km = (mask.reshape(mask.shape[0], mask.shape[1], -1).sum(-1) > 0).numpy().astype(np.uint8)
    return np.packbits(km.reshape(-1))


def sha_of(arr: np.ndarray) -> str:
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def synthetic_cases(folder):
    """Three small preprocessed 'cases' in the reference's on-disk form ([modalities..., seg] as <name>.npy next to the
    <name>.npz path the dataset dict names, properties with class_locations), from closed forms: the fixture of the
    DataLoader3D golden (tools/make_golden.py dataloader) and of its test.  The second case is smaller than the patch."""
    from collections import OrderedDict
    shapes = {"case_a": (18, 22, 30), "case_b": (10, 16, 12), "case_c": (25, 20, 21)}
    dataset = OrderedDict()
    for ci, (name, shp) in enumerate(shapes.items()):
        n = int(np.prod(shp))
        j = np.arange(n, dtype=np.float64).reshape(shp)
        mods = [np.sin(0.37 * j + 1.1 * ci + m).astype(np.float32) * (1 + m) for m in range(2)]
        zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
        seg = ((zz * 3 + yy * 5 + xx * 7 + ci) % 11 < 2).astype(np.float32) + ((zz + yy + xx) % 13 == 0).astype(np.float32) * 2
        seg = np.minimum(seg, 2.0)
        if ci == 2:
            seg[seg == 2] = 0                       # a case without class 2
        arr = np.stack(mods + [seg]).astype(np.float32)
        np.save(os.path.join(folder, name + ".npy"), arr)
        locs = OrderedDict((c, np.argwhere(seg == c)) for c in (1, 2))
        dataset[name] = OrderedDict(data_file=os.path.join(folder, name + ".npz"),
                                    properties=OrderedDict(class_locations=locs, name=name))
    return dataset


# ---------------------------------------------------------------------------------- gradients under the engine's own branch decisions
def engine_branches(eng):
    """The LeakyReLU masks and pooling arg-maxes the ENGINE took in its last forward, as oracle.Branches: every consumer forms
    u = fma(y, scale, shift) from the stored pre-norm output (e2e::in_act), whose sign the fp64 product-and-sum has exactly; the
    pooling kernel keeps the first maximum in (d, h, w) order like ATen."""
    import oracle
    import torch.nn.functional as F
    from e2enet_medical_amd.engine import ConvOp, PoolOp
    br = oracle.Branches()

    def pre(a):
        B, C = a.shape[:2]
        return a.data.double() * a.scale.double().view(B, C, 1, 1, 1) + a.shift.double().view(B, C, 1, 1, 1)
    for op in eng.ops:
        if isinstance(op, ConvOp):
            br.lrelu[op.prefix] = (pre(op.out) > 0).cpu()
        elif isinstance(op, PoolOp) and op.src.normed:
            u = pre(op.src).float()
            v = torch.where(u > 0, u, u * 0.01)
            br.pool[op.src.name] = F.max_pool3d(v, op.kernel, return_indices=True)[1].cpu()
    return br


def check_grads_same_branches(eng, spec, params, x, targets, w, shapes, tol_global=1e-4, tol_tensor=3e-4, factor=3.0):
    """The sharp gradient check: the fp64 oracle evaluated WITH the engine's own LeakyReLU / pooling decisions is a smooth function
    the engine's backward pass differentiates too, so the two gradients differ by rounding only -- against the per cents any two
    plain evaluations differ by (_check_all_grads).  The fp32 CPU oracle is put through the same decisions as the yardstick of
    what fp32 rounding costs on this graph (InstanceNorms over 8 voxels at the deep levels amplify it): the engine must stay within
    3x of it or within the absolute bars (1e-4 global, 3e-4 per tensor), per tensor (relative L2) and over all gradients together.
    Measured (round 5): 64^3 nets (config 5 at both densities, width 48) engine 3.2-4.3e-5 global against 3.6-7.7e-5 for the CPU
    path; config 1 (5 x 7 planes at the deep levels) 4.8-6.2e-4 against 2.7-2.8e-4.  This check found the one place where the
    engine's backward re-derived a branch decision instead of repeating the forward's (e2e_in_lrelu_bwd, ABI 16).  Tensors whose exact gradient
    is zero (conv biases in front of an InstanceNorm): max norm against the scale of the other gradients, as _check_all_grads."""
    import oracle
    br = engine_branches(eng)

    def forced(dtype):
        leaves = {n: p.detach().to(dtype).clone().requires_grad_(True) for n, p in params.items()}
        ref = oracle.forward(spec, leaves, x.to(dtype), branches=br)
        oracle.deep_supervision_loss(ref, targets, w, False).backward()
        return {n: leaves[n].grad.double() for n in shapes}
    g64, g32 = forced(torch.float64), forced(torch.float32)
    worst = {"engine": (0.0, None), "cpu32": (0.0, None)}
    num = {"engine": 0.0, "cpu32": 0.0}
    den = 0.0
    gscale = max(1.0, max(float(g64[n].abs().max()) for n in shapes))      # (networks with large activations have large gradients)
    for n in shapes:
        r = g64[n]
        den += r.pow(2).sum().item()
        for who, got in (("engine", eng.grads[n].cpu().double()), ("cpu32", g32[n])):
            d = got - r
            num[who] += d.pow(2).sum().item()
            if r.norm().item() > 1e-6:
                e = d.norm().item() / r.norm().item()
                if e > worst[who][0]:
                    worst[who] = (e, n)
            elif who == "engine":
                assert d.abs().max().item() <= 2e-3 * gscale, (n, "zero-gradient tensor", d.abs().max().item(), gscale)
    glob = {k: (v / den) ** 0.5 for k, v in num.items()}
    print("[grad, same branches] vs fp64: engine global rel-L2 %.3e worst tensor %.3e (%s) | cpu32 global %.3e worst %.3e (%s)"
          % (glob["engine"], worst["engine"][0], worst["engine"][1], glob["cpu32"], worst["cpu32"][0], worst["cpu32"][1]))
    assert glob["engine"] <= max(tol_global, factor * glob["cpu32"]), ("global", glob)
    assert worst["engine"][0] <= max(tol_tensor, factor * worst["cpu32"][0]), ("worst tensor", worst)
    return glob, worst


def write_synthetic_task(pre_root, task="Task998_Synth", plans=None, n_cases=6, shape=(20, 44, 40), mods=1, with_gt=True):
    """A preprocessed nnU-Net task folder in the reference's on-disk form, from closed forms:
      <pre_root>/<task>/nnUNetPlansv2.1_plans_3D.pkl, <task>/<data_identifier>_stage0/<case>.npy ([modalities..., seg]) + <case>.pkl
      (properties: list_of_data_files, crop box inside a larger original volume, spacings, class_locations; the last case was
      'resampled': its size_after_cropping differs from the stored grid), <task>/gt_segmentations/<case>.npy (labels on the ORIGINAL
      grid; .npy because SimpleITK is absent from the image).  Returns (dataset_directory, plans)."""
    import pickle
    from collections import OrderedDict
    plans = dict(plans)
    plans.setdefault('data_identifier', "nnUNetData_plans_v2.1")
    ddir = os.path.join(pre_root, task)
    folder = os.path.join(ddir, plans['data_identifier'] + "_stage0")
    os.makedirs(folder, exist_ok=True)
    os.makedirs(os.path.join(ddir, "gt_segmentations"), exist_ok=True)
    with open(os.path.join(ddir, "nnUNetPlansv2.1_plans_3D.pkl"), "wb") as f:
        pickle.dump(plans, f)
    for ci in range(n_cases):
        shp = tuple(s + 2 * (ci % 3) for s in shape)
        j = np.arange(int(np.prod(shp)), dtype=np.float64).reshape(shp)
        data = [np.sin(0.21 * j + 0.7 * ci + m).astype(np.float32) * (1 + m) for m in range(mods)]
        zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
        seg = (((zz // 3 + yy // 5 + xx // 4 + ci) % 5) < 2).astype(np.float32) + (((zz + yy + xx) % 17) == 0).astype(np.float32)
        seg = np.minimum(seg, 2.0)
        name = "case_%02d" % ci
        np.save(os.path.join(folder, name + ".npy"), np.stack(data + [seg]).astype(np.float32))
        resampled = ci == n_cases - 1
        after_crop = tuple(int(round(s * 1.25)) for s in shp) if resampled else shp
        off = (1 + ci % 2, 2, 3)
        orig = tuple(a + o + 2 for a, o in zip(after_crop, off))
        props = OrderedDict(class_locations=OrderedDict((c, np.argwhere(seg == c)) for c in (1, 2)), name=name,
                            list_of_data_files=["/raw/%s_0000.nii.gz" % name], original_size_of_raw_data=np.array(orig),
                            crop_bbox=[[o, o + a] for o, a in zip(off, after_crop)], size_after_cropping=np.array(after_crop),
                            original_spacing=np.array([1.0, 1.0, 1.0]),
                            spacing_after_resampling=np.array([1.25, 1.25, 1.25] if resampled else [1.0, 1.0, 1.0]),
                            itk_spacing=(1.0, 1.0, 1.0), itk_origin=(0.0, 0.0, 0.0),
                            itk_direction=(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0))
        with open(os.path.join(folder, name + ".pkl"), "wb") as f:
            pickle.dump(props, f)
        if with_gt:
            gt = np.zeros(orig, dtype=np.uint8)
            if resampled:
                idx = [np.minimum((np.arange(a) / 1.25).astype(int), s - 1) for a, s in zip(after_crop, shp)]
                inner = seg[np.ix_(*idx)]
            else:
                inner = seg
            gt[tuple(slice(o, o + a) for o, a in zip(off, after_crop))] = inner.astype(np.uint8)
            np.save(os.path.join(ddir, "gt_segmentations", name + ".npy"), gt)
    return ddir, plans
