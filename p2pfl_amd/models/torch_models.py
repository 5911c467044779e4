"""Plain ``torch.nn`` definitions of the federated models (the fp32 oracle of the HIP engine and
the single-node compatibility modules handed to ``Node(model=...)``).

Constructor convention follows the reference (``Model(in_channels, out_channels,
learning_rate, metrics=None, confusion_matrix=None, seed=None)``,
fedstellar/learning/pytorch/mnist/models/mlp.py:108-116). No Lightning/torchmetrics
dependency: training, metrics and logging are owned by :mod:`p2pfl_amd.learning`.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import specs


class FederatedModule(nn.Module):
    """Base class: carries the :class:`ModelSpec` that the batched engine executes."""

    spec_factory = None

    def __init__(self, in_channels=1, out_channels=10, learning_rate=1e-3, metrics=None,
                 confusion_matrix=None, seed=None):
        super().__init__()
        if seed is not None:
            torch.manual_seed(seed)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.learning_rate = learning_rate
        self.criterion = nn.CrossEntropyLoss()
        self.example_input_array = torch.zeros(1, 1, 28, 28)

    is_classifier = True        # metrics are derived from a confusion matrix of argmax predictions

    @property
    def spec(self):
        """:class:`ModelSpec` for models the client-batched HIP engine can execute, else None."""
        if type(self).spec_factory is None:
            return None
        return type(self).spec_factory(num_classes=self.out_channels, lr=self.learning_rate)

    def training_loss(self, x, y):
        """(loss, predictions-or-None) of one batch; overridden by non-classification models."""
        out = self(x)
        return self.criterion(out, y), out.detach().argmax(dim=1)

    def configure_optimizers(self):
        if self.spec is None:
            return torch.optim.Adam(self.parameters(), lr=self.learning_rate)
        o = self.spec.optim
        if o.kind == "adam":
            return torch.optim.Adam(self.parameters(), lr=o.lr, betas=(o.beta1, o.beta2), eps=o.eps,
                                    weight_decay=o.weight_decay)
        return torch.optim.SGD(self.parameters(), lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay)


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


_BY_SPEC = {"mnist-mlp": MNISTModelMLP, "mnist-cnn": MNISTModelCNN, "femnist-cnn": FEMNISTModelCNN}


def module_for_spec(spec: specs.ModelSpec, seed=None) -> FederatedModule:
    cls = _BY_SPEC[spec.name]
    return cls(out_channels=spec.num_classes, learning_rate=spec.optim.lr, seed=seed)
