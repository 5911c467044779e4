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


# The three spec'd families are defined under the reference's module paths; they are resolved lazily here so that either
# module can be imported first.
_LAZY = {"MNISTModelMLP": "p2pfl_amd.learning.pytorch.mnist.models.mlp",
         "MNISTModelCNN": "p2pfl_amd.learning.pytorch.mnist.models.cnn",
         "FEMNISTModelCNN": "p2pfl_amd.learning.pytorch.femnist.models.cnn"}
_BY_SPEC = {"mnist-mlp": "MNISTModelMLP", "mnist-cnn": "MNISTModelCNN", "femnist-cnn": "FEMNISTModelCNN"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def module_for_spec(spec: specs.ModelSpec, seed=None) -> FederatedModule:
    cls = __getattr__(_BY_SPEC[spec.name])
    return cls(out_channels=spec.num_classes, learning_rate=spec.optim.lr, seed=seed)
