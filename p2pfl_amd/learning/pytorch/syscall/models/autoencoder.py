"""Syscall autoencoder (reference: fedstellar/learning/pytorch/syscall/models/autoencoder.py:16-218):
17 -> 64 -> 16 -> 8 -> 16 -> 64 -> 17, MSE reconstruction loss, Adam lr 1e-2. The reference
``forward`` returns its input (autoencoder.py:153-157), i.e. a constant zero loss; this one
returns the reconstruction."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models import specs
from .....models.torch_models import FederatedModule


class SyscallModelAutoencoder(FederatedModule):
    is_classifier = False
    spec_factory = staticmethod(specs.syscall_autoencoder)    # client-batched HIP kernels (loss: csrc/loss.hip, mse_input)

    def __init__(self, in_channels=17, out_channels=9, learning_rate=1e-2, metrics=None, confusion_matrix=None,
                 seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 17)
        self.criterion = nn.MSELoss()
        self.fc1, self.fc2, self.fc3 = nn.Linear(in_channels, 64), nn.Linear(64, 16), nn.Linear(16, 8)
        self.fc4, self.fc5, self.fc6 = nn.Linear(8, 16), nn.Linear(16, 64), nn.Linear(64, in_channels)

    def encode(self, x):
        return torch.relu(self.fc3(torch.relu(self.fc2(torch.relu(self.fc1(x))))))

    def decode(self, z):
        return self.fc6(torch.relu(self.fc5(torch.relu(self.fc4(z)))))

    def forward(self, x):
        return self.decode(self.encode(x))

    def training_loss(self, x, y):
        return self.criterion(self(x), x), None
