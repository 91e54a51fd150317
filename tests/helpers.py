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
