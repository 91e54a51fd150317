"""reference e2enet/training/loss_functions/crossentropy.py:4-12 (class kept for API parity; the training fast path
computes CE inside the fused HIP loss kernel, see dice_loss.DC_and_CE_loss)."""
from torch import nn, Tensor


class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    """target is float with an extra channel dimension"""

    def forward(self, input: Tensor, target: Tensor) -> Tensor:
        if len(target.shape) == len(input.shape):
            assert target.shape[1] == 1
            target = target[:, 0]
        return super().forward(input, target.long())
