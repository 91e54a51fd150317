"""Dice + cross-entropy loss on the fused HIP kernels.

Drop-in for reference e2enet/training/loss_functions/dice_loss.py ``DC_and_CE_loss`` (:302-359) in the configuration
the trainer builds (nnUNetTrainer_simple.py:100): SoftDiceLoss(softmax, batch_dice, do_bg=False, smooth=1e-5) +
RobustCrossEntropyLoss, aggregate="sum", unit weights.  Forward and backward are the two kernels
``e2e_dc_ce_reduce`` / ``e2e_dc_ce_grad``; the module is an autograd node so it also composes with the reference's
``MultipleOutputLoss2``.
"""
import torch
from torch import nn

from ..._lib import lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _DcCeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, batch_dice, smooth):
        if not logits.is_cuda:
            raise RuntimeError("DC_and_CE_loss (MI355X) needs GPU tensors: there is no CPU fallback")
        logits = logits.contiguous().float()
        target = target.contiguous().float()
        b, k = logits.shape[:2]
        spatial = logits[0, 0].numel()
        ws = torch.empty(lib().loss_ws_bytes(b, k) // 8, dtype=torch.float64, device=logits.device)
        dl = torch.empty_like(logits)
        loss = torch.zeros(1, dtype=torch.float32, device=logits.device)
        lib().dc_ce_reduce(logits.data_ptr(), target.data_ptr(), ws.data_ptr(), b, k, spatial, _stream())
        lib().dc_ce_grad(logits.data_ptr(), target.data_ptr(), ws.data_ptr(), 1.0, 1 if batch_dice else 0, float(smooth),
                         dl.data_ptr(), loss.data_ptr(), b, k, spatial, _stream())
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None


class DC_and_CE_loss(nn.Module):
    def __init__(self, soft_dice_kwargs, ce_kwargs, aggregate="sum", square_dice=False, weight_ce=1, weight_dice=1,
                 log_dice=False, ignore_label=None):
        super().__init__()
        if square_dice or log_dice or ignore_label is not None or weight_ce != 1 or weight_dice != 1 or aggregate != "sum" \
                or ce_kwargs:
            raise NotImplementedError("fused DC_and_CE_loss implements the trainer's configuration only "
                                      "(nnUNetTrainer_simple.py:100)")
        if soft_dice_kwargs.get('do_bg', True):
            raise NotImplementedError("fused DC_and_CE_loss implements do_bg=False")
        self.batch_dice = bool(soft_dice_kwargs.get('batch_dice', False))
        self.smooth = float(soft_dice_kwargs.get('smooth', 1.))

    def forward(self, net_output, target):
        return _DcCeFunction.apply(net_output, target, self.batch_dice, self.smooth)
