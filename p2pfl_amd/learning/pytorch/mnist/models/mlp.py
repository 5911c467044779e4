"""MNIST MLP (reference: fedstellar/learning/pytorch/mnist/models/mlp.py:28-222). The metric / optimizer plumbing every
reference model file repeats lives once in ``models/torch_models.py: FederatedModule``."""
import torch
import torch.nn as nn

from .....models import specs
from .....models.torch_models import FederatedModule


class MNISTModelMLP(FederatedModule):
    """784-256-128-C MLP, log-softmax output (reference mnist/models/mlp.py:144-162)."""
    spec_factory = staticmethod(specs.mnist_mlp)

    def __init__(self, in_channels=1, out_channels=10, learning_rate=1e-3, metrics=None,
                 confusion_matrix=None, seed=None):
        super().__init__(in_channels, out_channels, learning_rate, metrics, confusion_matrix, seed)
        self.l1 = nn.Linear(28 * 28, 256)
        self.l2 = nn.Linear(256, 128)
        self.l3 = nn.Linear(128, out_channels)

    def forward(self, x):
        x = x.reshape(x.shape[0], -1)
        x = torch.relu(self.l1(x))
        x = torch.relu(self.l2(x))
        return torch.log_softmax(self.l3(x), dim=1)
