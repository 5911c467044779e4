"""One-class linear SVM on syscall features (reference: fedstellar/learning/pytorch/syscall/models/svm.py:16-211):
parameters ``w[17]`` and ``rho[1]``, loss ``0.5 |w|^2 + nu * mean(relu(1 - (x w - rho)))``, SGD lr 1e-2."""
from __future__ import annotations

import torch
import torch.nn as nn

from .....models.torch_models import FederatedModule


class SyscallModelSGDOneClassSVM(FederatedModule):
    is_classifier = False

    def __init__(self, in_channels=17, out_channels=9, learning_rate=1e-2, metrics=None, confusion_matrix=None,
                 seed=None, nu=0.1):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.example_input_array = torch.rand(1, 17)
        self.nu = nu
        self.w = nn.Parameter(torch.zeros(in_channels))
        self.rho = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return torch.matmul(x, self.w) - self.rho

    def hinge_loss(self, y):
        return torch.mean(torch.clamp(1 - y, min=0))

    def training_loss(self, x, y):
        return 0.5 * torch.sum(self.w ** 2) + self.nu * self.hinge_loss(self(x)), None

    def configure_optimizers(self):
        return torch.optim.SGD(self.parameters(), lr=self.learning_rate)
