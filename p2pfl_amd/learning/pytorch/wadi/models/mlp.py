"""WADI MLP (reference: fedstellar/learning/pytorch/wadi/models/mlp.py:16-228):
123 -> 1024 -> 512 -> 256 -> 128 -> 64 -> 32 -> 16 -> 8 -> out, ReLU between, sigmoid + BCE, Adam lr 1e-3."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models import specs
from .....models.torch_models import FederatedModule


class WADIModelMLP(FederatedModule):
    spec_factory = staticmethod(specs.wadi_mlp)     # runs on the client-batched HIP kernels (loss: csrc/loss.hip, C = 1)

    def __init__(self, in_channels=123, out_channels=1, learning_rate=1e-3, metrics=None, confusion_matrix=None,
                 seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.zeros(1, 123)
        self.criterion = nn.BCELoss()
        dims = [in_channels, 1024, 512, 256, 128, 64, 32, 16, 8, out_channels]
        for i in range(9):
            setattr(self, f"l{i + 1}", nn.Linear(dims[i], dims[i + 1]))

    def forward(self, x):
        for i in range(1, 9):
            x = torch.relu(getattr(self, f"l{i}")(x))
        return torch.sigmoid(self.l9(x))

    def training_loss(self, x, y):
        p = self(x)
        return self.criterion(p, y.unsqueeze(1).float()), (p.detach().squeeze(1) > 0.5).long()
