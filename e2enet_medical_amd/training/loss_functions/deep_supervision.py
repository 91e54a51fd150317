"""Deep-supervision wrapper (reference e2enet/training/loss_functions/deep_supervision.py:19-43)."""
from torch import nn


class MultipleOutputLoss2(nn.Module):
    def __init__(self, loss, weight_factors=None):
        super().__init__()
        self.weight_factors = weight_factors
        self.loss = loss

    def forward(self, x, y):
        assert isinstance(x, (tuple, list)), "x must be either tuple or list"
        assert isinstance(y, (tuple, list)), "y must be either tuple or list"
        weights = [1] * len(x) if self.weight_factors is None else self.weight_factors
        total = weights[0] * self.loss(x[0], y[0])
        for i in range(1, len(x)):
            if weights[i] != 0:
                total = total + weights[i] * self.loss(x[i], y[i])
        return total
