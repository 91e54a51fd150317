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
