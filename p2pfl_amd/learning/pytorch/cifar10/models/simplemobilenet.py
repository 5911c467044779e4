"""SimpleMobileNetV1 for CIFAR-10 (reference: fedstellar/learning/pytorch/cifar10/models/simplemobilenet.py:95-233):
stem 32 + depthwise-separable 64, 128(s2), 128, 256(s2), 256 + GAP + fc(256->C), Adam lr 1e-3."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models.torch_models import FederatedModule
from .fastermobilenet import conv_dw


class SimpleMobileNetV1(FederatedModule):
    def __init__(self, in_channels=3, out_channels=10, learning_rate=1e-3, metrics=None, confusion_matrix=None,
                 seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 3, 32, 32)
        self.model = nn.Sequential(
            nn.Conv2d(3, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            conv_dw(32, 64, 1), conv_dw(64, 128, 2), conv_dw(128, 128, 1), conv_dw(128, 256, 2), conv_dw(256, 256, 1),
            nn.AdaptiveAvgPool2d(1))
        self.fc = nn.Linear(256, out_channels)

    def forward(self, x):
        x = self.model(x.reshape(-1, 3, 32, 32))
        return self.fc(x.view(-1, 256))


SimpleMobileNet = SimpleMobileNetV1
