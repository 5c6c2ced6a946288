"""dun_amd: MI355X-native hot path of Depth Uncertainty Networks behind the reference's Python API.

Mirrors, name for name, what the reference's callers import on this path:
    src.probability                 -> dun_amd.probability
    src.DUN.stochastic_fc_models    -> dun_amd.stochastic_fc_models
    src.DUN.stochastic_img_resnets  -> dun_amd.stochastic_img_resnets
    src.DUN.training_wrappers       -> dun_amd.training_wrappers
    src.DUN.train_fc                -> dun_amd.train_fc
    src.utils (BaseNet, ...)        -> dun_amd.utils
The arithmetic runs in hand-written HIP kernels (dun_amd/csrc) reached through the C ABI of
include/dun_hip.h; there is no CPU implementation in this package.
"""
__version__ = "0.1.0"
