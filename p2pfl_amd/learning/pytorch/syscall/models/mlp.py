"""Syscall MLP (reference: fedstellar/learning/pytorch/syscall/models/mlp.py:16-213):
17 -> 256 -> BN -> ReLU -> Dropout(.5) -> 128 -> BN -> ReLU -> Dropout -> 9 -> log_softmax, Adam lr 1e-2."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models.torch_models import FederatedModule


class SyscallModelMLP(FederatedModule):
    def __init__(self, in_channels=17, out_channels=9, learning_rate=1e-2, metrics=None, confusion_matrix=None,
                 seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 17)
        self.l1 = nn.Linear(in_channels, 256)
        self.batchnorm1 = nn.BatchNorm1d(256)
        self.dropout = nn.Dropout(0.5)
        self.l2 = nn.Linear(256, 128)
        self.batchnorm2 = nn.BatchNorm1d(128)
        self.l3 = nn.Linear(128, out_channels)

    def forward(self, x):
        x = self.dropout(torch.relu(self.batchnorm1(self.l1(x))))
        x = self.dropout(torch.relu(self.batchnorm2(self.l2(x))))
        return torch.log_softmax(self.l3(x), dim=1)
