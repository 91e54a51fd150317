"""Ablation network ``unetpp_d_331`` of the reference (e2enet/network_architecture/unetpp_d_331.py: conv kernel (3,3,1), the
depth shift switched off in its source, :102) on the MI355X engine.  Same module path and class name as the reference, so
``from ...unetpp_d_331 import Generic_UNetPlusPlus`` (nnUNetTrainer_simple.py:315) keeps working; the implementation is
the (1,3,3) engine on axis-permuted tensors (see ``unetpp_d.Generic_UNetPlusPlus``, ``conv_variant``)."""
from .unetpp_d import (Generic_UNetPlusPlus as _Base, ConvDropoutNormNonlin, StackedConvLayers,   # noqa: F401
                       InitWeights_He, softmax_helper)


class Generic_UNetPlusPlus(_Base):
    def __init__(self, *args, **kwargs):
        kwargs["conv_variant"] = "331"
        super().__init__(*args, **kwargs)
