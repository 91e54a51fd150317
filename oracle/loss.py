"""Deep-supervision Dice + CE loss and the hard Dice metric (oracle; test infrastructure only).

Restates reference
  e2enet/training/loss_functions/dice_loss.py:100-153 (get_tp_fp_fn_tn),
  :156-192 (SoftDiceLoss), :302-359 (DC_and_CE_loss),
  e2enet/training/loss_functions/crossentropy.py:4-12,
  e2enet/training/loss_functions/deep_supervision.py:31-43 (MultipleOutputLoss2),
  nnUNetTrainer_simple.py:200-213 (deep-supervision weights), :100 (loss kwargs),
  e2enet/evaluation/metrics.py:106-121 (hard Dice = 2TP / (2TP + FP + FN)).
"""
import numpy as np
import torch
import torch.nn.functional as F


def ds_weights(net_numpool: int):
    w = np.array([1 / (2 ** i) for i in range(net_numpool)])
    keep = np.array([True] + [i < net_numpool - 1 for i in range(1, net_numpool)])
    w[~keep] = 0
    return w / w.sum()


def dc_ce_loss(logits: torch.Tensor, target: torch.Tensor, batch_dice=False, smooth=1e-5):
    """logits [B,K,...], target [B,1,...] float labels.  do_bg=False, weight_ce=weight_dice=1."""
    p = F.softmax(logits, 1)
    gt = target.long()
    onehot = torch.zeros(p.shape, device=p.device)
    onehot.scatter_(1, gt, 1)
    axes = ([0] if batch_dice else []) + list(range(2, p.dim()))
    tp = (p * onehot).sum(axes)
    fp = (p * (1 - onehot)).sum(axes)
    fn = ((1 - p) * onehot).sum(axes)
    dc = (2 * tp + smooth) / (2 * tp + fp + fn + smooth + 1e-8)
    dc = dc[1:] if batch_dice else dc[:, 1:]
    dice_term = -dc.mean()
    ce = F.cross_entropy(logits, target[:, 0].long())
    return ce + dice_term


def deep_supervision_loss(outputs, targets, weights, batch_dice=False):
    total = weights[0] * dc_ce_loss(outputs[0], targets[0], batch_dice)
    for i in range(1, len(outputs)):
        if weights[i] != 0:
            total = total + weights[i] * dc_ce_loss(outputs[i], targets[i], batch_dice)
    return total


def hard_dice(test: np.ndarray, reference: np.ndarray, label=None) -> float:
    """metrics.py:106-121 on boolean maps (``label`` selects a class from label maps)."""
    if label is not None:
        test = test == label
        reference = reference == label
    test = test != 0
    reference = reference != 0
    tp = int((test & reference).sum())
    fp = int((test & ~reference).sum())
    fn = int((~test & reference).sum())
    if tp + fp + fn == 0:
        return float("nan")
    return float(2. * tp / (2 * tp + fp + fn))
