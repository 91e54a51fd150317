"""He initialisation (reference e2enet/network_architecture/initialization.py:17-26, duplicated at unetpp_d.py:28-36)."""
from torch import nn


class InitWeights_He(object):
    """kaiming_normal_(a=neg_slope) on every (transposed) conv weight, zero bias."""

    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module):
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)
