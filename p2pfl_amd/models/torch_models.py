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

    # ------------------------------------------------------------------ step / metric hooks of the reference models
    # The reference's model files are LightningModules that each repeat step / training_step / validation_step /
    # test_step, process_metrics (torchmetrics of the batch, logged per step), log_metrics_by_epoch (epoch metrics +
    # confusion matrix) and the on_*_epoch_end hooks (mnist/models/mlp.py:33-222). They live once here, on a confusion
    # matrix per phase (learning/metrics.py); ``log`` / ``log_dict`` go to ``self.logger`` when one is attached and to
    # ``self.logged_metrics`` always. The batched engines do not call these (their loss, counters and optimizer are
    # kernels); code written against the reference's models - custom loops, subclasses overriding a hook - does.
    PHASES = ("Train", "Validation", "Test")

    @property
    def device(self):
        for p_ in self.parameters():
            return p_.device
        return torch.device("cpu")

    def _metric_state(self):
        st = self.__dict__.get("_fed_metric_state")
        if st is None:
            C = max(int(self.out_channels), 2)
            st = {"cm": {ph: torch.zeros(C, C, dtype=torch.int64) for ph in self.PHASES},
                  "epoch": {ph: 0 for ph in self.PHASES}, "logged": {}}
            self.__dict__["_fed_metric_state"] = st
        return st

    @property
    def epoch_global_number(self):
        return self._metric_state()["epoch"]

    @property
    def logged_metrics(self):
        """Last value of every metric logged through :meth:`log` / :meth:`log_dict`."""
        return self._metric_state()["logged"]

    def log(self, name, value, **kwargs):
        self.log_dict({name: value}, **kwargs)

    def log_dict(self, metrics, **kwargs):
        vals = {k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in metrics.items()}
        self._metric_state()["logged"].update(vals)
        lg = self.__dict__.get("logger")
        if lg is not None:
            lg.log_metrics(vals)

    def _phase_metrics(self, cm):
        from ..learning.metrics import metrics_from_confusion
        m = metrics_from_confusion(cm)
        m.pop("MicroAccuracy", None)
        return m

    def process_metrics(self, phase, y_pred, y, loss=None):
        """Loss and macro Accuracy / Precision / Recall / F1Score of ONE batch as ``<phase>/<name>``; the batch also goes
        into the phase's confusion matrix for :meth:`log_metrics_by_epoch`."""
        if phase not in self.PHASES:
            raise NotImplementedError(phase)
        if loss is not None:
            self.log(f"{phase}/Loss", loss)
        if not self.is_classifier or y_pred is None:
            return
        C = self._metric_state()["cm"][phase].shape[0]
        pred = y_pred.argmax(dim=1) if y_pred.dim() > 1 else y_pred
        idx = (y.long().cpu().clamp(0, C - 1) * C + pred.long().cpu().clamp(0, C - 1)).reshape(-1)
        bcm = torch.zeros(C * C, dtype=torch.int64).index_add_(0, idx, torch.ones_like(idx)).view(C, C)
        self.log_dict({f"{phase}/{k}": v for k, v in self._phase_metrics(bcm).items()})
        self._metric_state()["cm"][phase] += bcm

    def log_metrics_by_epoch(self, phase, print_cm=False, plot_cm=False):
        """Epoch metrics of ``phase`` as ``<phase>Epoch/<name>`` from the accumulated confusion matrix, which is then
        reset; ``print_cm`` prints it, ``plot_cm`` hands it to the logger as a table (``log_table``) where it has one."""
        if phase not in self.PHASES:
            raise NotImplementedError(phase)
        st = self._metric_state()
        cm = st["cm"][phase]
        if self.is_classifier:
            self.log_dict({f"{phase}Epoch/{k}": v for k, v in self._phase_metrics(cm).items()})
            if print_cm:
                print(f"{phase}Epoch/CM\n", cm)
            lg = self.__dict__.get("logger")
            if plot_cm and lg is not None and hasattr(lg, "log_table"):
                lg.log_table(f"{phase}Epoch/CM", columns=[str(i) for i in range(cm.shape[1])], data=cm.tolist(),
                             step=st["epoch"][phase])
        st["cm"][phase] = torch.zeros_like(cm)
        st["epoch"][phase] += 1

    def step(self, batch, phase):
        x, y = batch
        x, y = x.to(self.device), y.to(self.device)
        if self.is_classifier:
            out = self(x)
            loss = self.criterion(out, y)
            self.process_metrics(phase, out.detach(), y, loss.detach())
        else:
            loss, _ = self.training_loss(x, y)
            self.process_metrics(phase, None, y, loss.detach())
        return loss

    def training_step(self, batch, batch_id=0):
        return self.step(batch, "Train")

    def validation_step(self, batch, batch_idx=0):
        return self.step(batch, "Validation")

    def test_step(self, batch, batch_idx=0):
        return self.step(batch, "Test")

    def on_train_epoch_end(self):
        self.log_metrics_by_epoch("Train", print_cm=False, plot_cm=True)

    def on_validation_epoch_end(self):
        self.log_metrics_by_epoch("Validation", print_cm=False, plot_cm=True)

    def on_test_epoch_end(self):
        self.log_metrics_by_epoch("Test", print_cm=False, plot_cm=True)

    def configure_optimizers(self):
        if self.spec is None:
            return torch.optim.Adam(self.parameters(), lr=self.learning_rate)
        o = self.spec.optim
        if o.kind == "adam":
            return torch.optim.Adam(self.parameters(), lr=o.lr, betas=(o.beta1, o.beta2), eps=o.eps,
                                    weight_decay=o.weight_decay)
        return torch.optim.SGD(self.parameters(), lr=o.lr, momentum=o.momentum, weight_decay=o.weight_decay)


# The spec'd families are defined under the reference's module paths; they are resolved lazily here so that either
# module can be imported first.
_LAZY = {"MNISTModelMLP": "p2pfl_amd.learning.pytorch.mnist.models.mlp",
         "MNISTModelCNN": "p2pfl_amd.learning.pytorch.mnist.models.cnn",
         "FEMNISTModelCNN": "p2pfl_amd.learning.pytorch.femnist.models.cnn",
         "WADIModelMLP": "p2pfl_amd.learning.pytorch.wadi.models.mlp",
         "SyscallModelAutoencoder": "p2pfl_amd.learning.pytorch.syscall.models.autoencoder"}
_BY_SPEC = {"mnist-mlp": "MNISTModelMLP", "mnist-cnn": "MNISTModelCNN", "femnist-cnn": "FEMNISTModelCNN",
            "wadi-mlp": "WADIModelMLP", "syscall-autoencoder": "SyscallModelAutoencoder"}


def __getattr__(name):
    if name in _LAZY:
        import importlib
        return getattr(importlib.import_module(_LAZY[name]), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def module_for_spec(spec: specs.ModelSpec, seed=None) -> FederatedModule:
    cls = __getattr__(_BY_SPEC[spec.name])
    if spec.name == "wadi-mlp":                     # one output, two classes (see specs.wadi_mlp)
        return cls(learning_rate=spec.optim.lr, seed=seed)
    return cls(out_channels=spec.num_classes, learning_rate=spec.optim.lr, seed=seed)
