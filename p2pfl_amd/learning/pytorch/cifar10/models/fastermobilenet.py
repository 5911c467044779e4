"""FasterMobileNet for CIFAR-10 (reference: fedstellar/learning/pytorch/cifar10/models/fastermobilenet.py:17-224):
stem 16 + depthwise-separable (16->32,s1) (32->64,s2) (64->64,s1) + GAP + fc(64->C), Adam lr 1e-3.
Same module names (``model.*``, ``fc``) so state_dicts interoperate."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models.torch_models import FederatedModule


def conv_dw(cin, cout, stride):
    return nn.Sequential(
        nn.Conv2d(cin, cin, 3, stride, 1, groups=cin, bias=False), nn.BatchNorm2d(cin), nn.ReLU(inplace=True),
        nn.Conv2d(cin, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class FasterMobileNet(FederatedModule):
    def __init__(self, in_channels=3, out_channels=10, learning_rate=1e-3, metrics=None, confusion_matrix=None,
                 seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 3, 32, 32)
        self.model = nn.Sequential(
            nn.Conv2d(3, 16, 3, 1, 1, bias=False), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
            conv_dw(16, 32, 1), conv_dw(32, 64, 2), conv_dw(64, 64, 1), nn.AdaptiveAvgPool2d(1))
        self.fc = nn.Linear(64, out_channels)

    def forward(self, x):
        x = self.model(x.reshape(-1, 3, 32, 32))
        return self.fc(x.view(-1, 64))
