"""FEMNIST CNN (reference: fedstellar/learning/pytorch/femnist/models/cnn.py:18-212)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from .....models import specs
from .....models.torch_models import FederatedModule


class FEMNISTModelCNN(FederatedModule):
    """conv7x7(1->32)-pool-conv3x3(32->64)-pool-fc C, SGD+momentum
    (reference femnist/models/cnn.py:136-157)."""
    spec_factory = staticmethod(specs.femnist_cnn)

    def __init__(self, in_channels=1, out_channels=62, learning_rate=1e-3, metrics=None,
                 confusion_matrix=None, seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.conv1 = nn.Conv2d(in_channels, 32, kernel_size=7, padding=3)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.out = nn.Linear(7 * 7 * 64, out_channels)

    def forward(self, x):
        x = x.reshape(-1, 1, 28, 28)
        x = F.max_pool2d(torch.relu(self.conv1(x)), 2, 2)
        x = F.max_pool2d(torch.relu(self.conv2(x)), 2, 2)
        x = x.reshape(x.shape[0], -1)
        return self.out(x)
