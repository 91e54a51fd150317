"""e2enet_medical_amd: MI355X-native engine for the E2ENet shiftConvPP + DSFF hot path.

Drop-in surface (same module paths below this package as below ``e2enet`` in the reference):
  network_architecture.unetpp_d.Generic_UNetPlusPlus, network_architecture.neural_network.SegmentationNetwork,
  training.network_training.sparselearning.core_channel.{Masking, CosineDecay, add_sparse_args},
  training.network_training.nnUNetTrainer_simple.nnUNetTrainer_simple, training.loss_functions.*.
All arithmetic on the hot path runs in hand-written HIP kernels (csrc/, C ABI in include/e2e_hip.h); there is no
CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
