"""MNIST CNN (reference: fedstellar/learning/pytorch/mnist/models/cnn.py:18-219; forward fixed to
return the logits)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from .....models import specs
from .....models.torch_models import FederatedModule


class MNISTModelCNN(FederatedModule):
    """conv5x5(1->32)-pool-conv5x5(32->64)-pool-fc2048-fc C (reference mnist/models/cnn.py:134-158).

    The reference ``forward`` returns its input (cnn.py:159); this one returns the logits.
    """
    spec_factory = staticmethod(specs.mnist_cnn)

    def __init__(self, in_channels=1, out_channels=10, learning_rate=1e-3, metrics=None,
                 confusion_matrix=None, seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.conv1 = nn.Conv2d(in_channels, 32, kernel_size=5, padding=2)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=5, padding=2)
        self.l1 = nn.Linear(7 * 7 * 64, 2048)
        self.l2 = nn.Linear(2048, out_channels)

    def forward(self, x):
        x = x.reshape(-1, 1, 28, 28)
        x = F.max_pool2d(torch.relu(self.conv1(x)), 2, 2)
        x = F.max_pool2d(torch.relu(self.conv2(x)), 2, 2)
        x = x.reshape(x.shape[0], -1)
        x = torch.relu(self.l1(x))
        return self.l2(x)
