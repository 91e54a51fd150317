"""Dynamic Sparse Feature Fusion masks (oracle; test infrastructure only).

Restates reference
e2enet/training/network_training/sparselearning/core_channel.py:
  * CosineDecay                        :32-41  (torch CosineAnnealingLR, recursive form)
  * Masking.init('uniform')            :141-169
  * Masking.apply_mask                 :427-434
  * Masking.step / truncate_weights    :290-317, :556-611
  * kernel_death                       :647-666
  * kernel_growth                      :721-739
  * kernel_grad_growth                 :771-790
All index draws use Python's ``random`` module exactly like the reference, so
with the same ``random.seed`` the mask indices are bit-identical.
"""
import math
import random
from typing import Dict
import numpy as np
import torch


class CosineDeathRate:
    """core_channel.py:32-41: death rate driven by torch's CosineAnnealingLR on a dummy SGD.
    The recursive update is kept (it differs from the closed form in the last ulp)."""

    def __init__(self, death_rate, t_max, eta_min=0.001):
        self._sgd = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=death_rate)
        self._sched = torch.optim.lr_scheduler.CosineAnnealingLR(self._sgd, t_max, eta_min, -1)

    def step(self):
        self._sched.step()

    def get_dr(self):
        return self._sgd.param_groups[0]["lr"]


def kernel_l1(weight: torch.Tensor) -> torch.Tensor:
    """Three chained last-dim sums of |w| (core_channel.py:652-655). The association order
    ((a+b)+c per row, then rows, then depth) is part of the bit-exact contract."""
    s = torch.sum(torch.abs(weight), dim=-1)
    s = torch.sum(s, dim=-1)
    return torch.sum(s, dim=-1)


def uniform_kernel_masks(shapes: "Dict[str, tuple]", names, density: float):
    """Masking.init(mode='uniform') (core_channel.py:141-169). Consumes ``random`` in
    ``names`` order (the reference iterates module.named_parameters())."""
    masks = {}
    for name in names:
        shp = tuple(shapes[name])
        dens = 0.2 if shp[0] == 48 else density         # quirk at :147-151
        k_size = int(np.prod(shp[-3:]))
        numel = int(np.prod(shp))
        kernel_num = round(numel * dens / k_size)
        picks = random.sample(list(range(0, shp[0] * shp[1])), kernel_num)
        m = torch.zeros(shp, dtype=torch.float32)
        if kernel_num:
            idx = torch.tensor(picks, dtype=torch.long)
            m[idx // shp[1], idx % shp[1]] = 1.0
        masks[name] = m
    return masks


def kernel_death(mask: torch.Tensor, weight: torch.Tensor, death_rate: float):
    """core_channel.py:647-666. Returns (new_mask (in place), prune_num)."""
    k_size = int(np.prod(weight.shape[-3:]))
    nonzeros = mask.sum().item()
    zeros = mask.numel() - nonzeros
    score = kernel_l1(weight)
    prune_num = math.ceil(death_rate * nonzeros / k_size)
    value, _ = torch.sort(score.reshape(-1))
    num_zeros = math.ceil(zeros / k_size)
    thr = value[num_zeros + prune_num - 1].item()
    dead = torch.nonzero(score <= thr)
    mask[dead[:, 0], dead[:, 1]] = 0.0
    return mask, prune_num


def kernel_growth(mask: torch.Tensor, num_growth: int):
    """core_channel.py:721-739: candidates = kernels whose mask sum < 1 in row-major order,
    ``random.sample`` picks which to revive."""
    new_mask = mask.to(torch.uint8).clone()
    s = new_mask.sum(dim=-1).sum(dim=-1).sum(dim=-1)
    cand = torch.nonzero(s < 1)
    picks = random.sample(list(range(0, cand.shape[0])), num_growth)
    g = cand[picks]
    new_mask[g[:, 0], g[:, 1]] = 1
    return new_mask.float()


def kernel_grad_growth(mask: torch.Tensor, grad: torch.Tensor, num_growth: int):
    """core_channel.py:771-790 (growth_mode='gradient'): per kernel -- and per depth slice of the kernel, only the LAST TWO
    axes are summed -- the sum of |weight.grad| where the mask is dead, 0 elsewhere; every entry strictly above the
    (num_growth)-th largest (0-based) revives its whole kernel."""
    new_mask = mask.to(torch.uint8).clone()
    if num_growth == 0:
        return new_mask.float()
    mask_sum = torch.squeeze(torch.sum(torch.sum(torch.abs(new_mask), dim=-1), dim=-1))
    data_sum = torch.squeeze(torch.sum(torch.sum(torch.abs(grad), dim=-1), dim=-1))
    score = data_sum * (mask_sum < 1).float()
    value, _ = torch.sort(score.reshape(-1), descending=True)
    idx = torch.nonzero(score > value[num_growth].item())
    new_mask[idx[:, 0], idx[:, 1]] = 1
    return new_mask.float()


class DsffState:
    """Host-side restatement of ``Masking`` restricted to death='magnitude', growth='random',
    redistribution='none' (the README/CLI defaults: core_channel.py:17-31)."""

    def __init__(self, params: "Dict[str, torch.Tensor]", names, density, death_rate,
                 t_max, update_frequency, momentum_buffers=None):
        self.params = params
        self.names = list(names)
        self.momentum = momentum_buffers if momentum_buffers is not None else {}
        self.decay = CosineDeathRate(death_rate, t_max)
        self.death_rate = death_rate
        self.update_frequency = update_frequency
        self.steps = 0
        shapes = {n: tuple(params[n].shape) for n in self.names}
        self.masks = uniform_kernel_masks(shapes, self.names, density)
        self.apply_mask()

    def apply_mask(self):
        for n in self.names:
            self.params[n].data = self.params[n].data * self.masks[n]
            if n in self.momentum:
                self.momentum[n] = self.momentum[n] * self.masks[n]

    def truncate_weights(self):
        num_death = {}
        for n in self.names:                                   # death pass (:558-581)
            self.masks[n], num_death[n] = kernel_death(self.masks[n], self.params[n].data,
                                                       self.death_rate)
        for n in self.names:                                   # growth pass (:583-609)
            self.masks[n] = kernel_growth(self.masks[n], num_death[n])
        self.apply_mask()

    def step(self):
        """core_channel.py:290-317."""
        self.apply_mask()
        self.decay.step()
        self.death_rate = self.decay.get_dr()
        self.steps += 1
        if self.update_frequency is not None and self.steps % self.update_frequency == 0:
            self.truncate_weights()
            return True
        return False
