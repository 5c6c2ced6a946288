"""Parameter containers for the skippable MLP blocks (reference: src/DUN/layers.py:33-53).

The modules keep the reference's attribute layout (``block = Sequential(Linear, ReLU, BatchNorm1d)``) so that
``state_dict()`` keys are interchangeable with the reference's.  Their ``forward`` is the torch-op
restatement used only by the cold ``vec_forward`` path (plots); the hot path never calls it -- the trunk
kernels read the same parameters through the model's flat buffer.
"""
import torch.nn as nn


class bern_MLPBlock(nn.Module):
    """x + b * BN(ReLU(Linear(x)))"""

    def __init__(self, width):
        super(bern_MLPBlock, self).__init__()
        self.block = nn.Sequential(nn.Linear(width, width), nn.ReLU(), nn.BatchNorm1d(num_features=width))

    def forward(self, x, b):
        return x + b * self.block(x)


class bern_MLPBlock_nores(nn.Module):
    """BN(ReLU(Linear(x))); the mask b is ignored, as in the reference"""

    def __init__(self, width):
        super(bern_MLPBlock_nores, self).__init__()
        self.block = nn.Sequential(nn.Linear(width, width), nn.ReLU(), nn.BatchNorm1d(num_features=width))

    def forward(self, x, b):
        return self.block(x)
