"""LightningLearner: the node-local learner (reference:
fedstellar/learning/pytorch/lightninglearner.py:38-193).

The name, constructor and the 17 ``NodeLearner`` methods are the reference's; PyTorch-Lightning
itself is not used (nor available on the target image). Training is owned by this class:

* models that carry a :class:`~p2pfl_amd.models.specs.ModelSpec` (MNIST MLP/CNN, FEMNIST CNN) on a
  GPU run through the gfx950 shard engine with ``K = 1`` (same kernels and hipGraph replay as the
  client-batched federation);
* every other model (CIFAR ResNets/MobileNets, syscall MLP/autoencoder/SVM, WADI MLP) and the CPU
  path use a plain autograd loop over the data module's loaders.

Kept reference behaviour: a fresh trainer — and thus fresh optimizer state — per round
(``fit`` re-creates it, lightninglearner.py:113-118), payload = ``pickle((arrays in state_dict order,
contributors, weight))`` (:73-89), ``check_parameters`` = key set + shapes (:91-99),
``get_num_samples`` = (train, test) sizes (:149-153), FL-aware logger stepping folded in
``finalize_round`` (:162-165), metric names ``{Train,Validation,Test}[Epoch]/{Loss,Accuracy,Precision,
Recall,F1Score}``.
"""
from __future__ import annotations

import logging
import threading
from collections import OrderedDict

import numpy as np
import torch

from ..codec import decode_parameters as _decode
from ..codec import encode_parameters as _encode
from ..exceptions import DecodingParamsError, ModelNotMatchingError
from ..learner import NodeLearner
from ..metrics import metrics_from_confusion


# Several nodes in ONE process (tests, examples/single_device.py) each own a one-client HipEngine and may fit from
# different threads at once: that needs no lock - the stall that once made engine calls take turns was the
# high-priority side stream (runtime/engine.py, HipEngine.__init__; tools/diag_stall.py).


def resolve_device(config) -> torch.device:
    acc = "auto"
    gpu = None
    if config is not None:
        da = config.participant.get("device_args", {})
        acc = str(da.get("accelerator", "auto")).lower()
        gpu = da.get("gpu")
    if acc in ("gpu", "cuda", "auto") and torch.cuda.is_available():
        return torch.device("cuda", int(gpu) if gpu is not None else torch.cuda.current_device())
    return torch.device("cpu")


class LightningLearner(NodeLearner):
    def __init__(self, model, data, config=None, logger=None):
        self.model = model
        self.data = data
        self.config = config
        self.logger = logger
        self.epochs = 1
        self.device = resolve_device(config)
        self._stop = threading.Event()
        self._engine = None
        self._optimizer = None
        self._trainer_fresh = False
        self.round = 0
        if self.logger is not None:
            self.logger.log_metrics({"Round": self.round}, step=getattr(self.logger, "global_step", 0))

    # ------------------------------------------------------------------ model / data
    def set_model(self, model):
        self.model = model
        self._engine = None

    def set_data(self, data):
        self.data = data
        self._engine = None

    # ------------------------------------------------------------------ payload
    def encode_parameters(self, params=None, contributors=None, weight=None):
        if params is None:
            params = self.model.state_dict()
        arrays = [v.detach().cpu().numpy() for v in params.values()]
        return _encode(arrays, contributors, weight)

    def decode_parameters(self, data):
        arrays, contributors, weight = _decode(data)
        keys = list(self.model.state_dict().keys())
        if len(arrays) != len(keys):
            raise DecodingParamsError(f"payload holds {len(arrays)} tensors, the model {len(keys)}")
        sd = OrderedDict((k, torch.tensor(np.asarray(a))) for k, a in zip(keys, arrays))
        return sd, contributors, weight

    def check_parameters(self, params):
        own = self.model.state_dict()
        if set(params.keys()) != set(own.keys()):
            return False
        return all(tuple(params[k].shape) == tuple(own[k].shape) for k in own)

    def set_parameters(self, params):
        try:
            self.model.load_state_dict(params)
        except Exception as e:
            raise ModelNotMatchingError(f"Not matching models: {e}") from e

    def get_parameters(self):
        return self.model.state_dict()

    # ------------------------------------------------------------------ trainer lifecycle
    def set_epochs(self, epochs):
        self.epochs = int(epochs)

    def init(self):
        self._stop.clear()

    def close(self):
        if self.logger is not None:
            try:
                self.logger.finalize("success")
            except Exception:
                pass

    def finalize_round(self):
        if self.logger is not None and hasattr(self.logger, "finalize_round"):
            self.logger.finalize_round()
        self.round += 1

    def create_trainer(self):
        """Fresh optimizer state (the reference builds a new Lightning Trainer per round)."""
        self._stop.clear()
        self._optimizer = None
        self._trainer_fresh = True

    def interrupt_fit(self):
        self._stop.set()

    def get_num_samples(self):
        return self.data.num_samples()

    # ------------------------------------------------------------------ engine plumbing
    def _use_engine(self) -> bool:
        return getattr(self.model, "spec", None) is not None and self.device.type == "cuda"

    def _get_engine(self):
        if self._engine is None:
            from ...runtime.engine import HipEngine
            bs = getattr(self.data, "batch_size", 32)
            eng = HipEngine(self.model.spec, 1, self.device, batch_size=bs, log_steps=True)
            un = lambda t: None if t is None else t.unsqueeze(0)
            eng.set_data(*(un(t) for t in self.data.tensors()))
            # The model's parameters BECOME views of slab row 0 (device-side pack once, then aliasing): what the
            # kernels train is what state_dict() / load_state_dict() see, with no copy at all per fit / evaluate -
            # the reference's D2H + pickle site (lightninglearner.py:73-89) is only touched when a payload is
            # really encoded for the wire.
            self.model.to(self.device)
            eng.layout.state_dict_to_row(self.model.state_dict(), eng.params[0])
            for name, p in self.model.named_parameters():
                p.data = eng.layout.view(eng.params, name)[0]
            self._engine = eng
        return self._engine

    def _model_to_engine(self, eng):
        """No-op when the parameters alias the slab row (always, unless someone re-assigned ``p.data``)."""
        for name, p in self.model.named_parameters():
            if p.data_ptr() != eng.layout.view(eng.params, name)[0].data_ptr():
                eng.layout.view(eng.params, name)[0].copy_(p.data.to(eng.params.device))
                p.data = eng.layout.view(eng.params, name)[0]

    def _engine_to_model(self, eng):
        pass            # the model's parameters are views of the row the engine just trained

    # ------------------------------------------------------------------ logging helpers
    def _log(self, metrics, step=None):
        if self.logger is not None and metrics:
            self.logger.log_metrics(metrics, step=step)

    def _log_phase(self, phase, loss, cm, step=None):
        out = {f"{phase}Epoch/Loss": float(loss), f"{phase}/Loss": float(loss)}
        if cm is not None:
            for k, v in metrics_from_confusion(cm).items():
                if k != "MicroAccuracy":
                    out[f"{phase}Epoch/{k}"] = v
        self._log(out, step)
        return out

    # ------------------------------------------------------------------ generic autograd loop
    _STEP_HOOK = {"Train": "training_step", "Validation": "validation_step", "Test": "test_step"}

    def _loss_and_pred(self, model, xb, yb, phase, i):
        """(loss, predictions-or-None) of a batch. Models of this package answer ``training_loss``; a model written for
        the reference's Trainer (a LightningModule-shaped class: ``training_step`` / ``validation_step`` / ``test_step``
        returning the loss or ``{"loss": ...}`` and logging through ``self.log``) is driven through those hooks."""
        if hasattr(model, "training_loss"):
            return model.training_loss(xb, yb)
        if not hasattr(model, "log"):          # plain nn.Module with Lightning-style hooks: give it somewhere to log
            num = lambda v: float(v.detach()) if torch.is_tensor(v) else float(v)
            model.log = lambda name, value, **kw: self._log({name: num(value)})
            model.log_dict = lambda d, **kw: self._log({k: num(v) for k, v in d.items()})
        out = getattr(model, self._STEP_HOOK[phase])((xb, yb), i)
        return (out["loss"] if isinstance(out, dict) else out), None

    def _run_loader(self, loader, phase, train):
        model, dev = self.model, self.device
        C = getattr(self.data, "num_classes", getattr(model, "out_channels", 2))
        C = max(C, 2)
        cm = torch.zeros(C, C, dtype=torch.int64) if getattr(model, "is_classifier", False) else None
        total, n, steps = 0.0, 0, 0
        if loader is None:
            return 0.0, cm, 0
        model.train(train)
        for xb, yb in loader:
            if self._stop.is_set():
                break
            xb, yb = xb.to(dev), yb.to(dev)
            if hasattr(self.data, "augment") and train:
                xb = self.data.augment(xb)                  # on the batch's device (no host round trip)
            if train:
                loss, pred = self._loss_and_pred(model, xb, yb, phase, steps)
                self._optimizer.zero_grad(set_to_none=True)
                loss.backward()
                self._optimizer.step()
            else:
                with torch.no_grad():
                    loss, pred = self._loss_and_pred(model, xb, yb, phase, steps)
            total += float(loss.detach()) * len(xb)
            n += len(xb)
            steps += 1
            idx = None
            if cm is not None and pred is not None:
                idx = (yb.long().cpu() * C + pred.long().cpu()).clamp(0, C * C - 1)
                cm.view(-1).index_add_(0, idx, torch.ones_like(idx))
            if train and steps % 20 == 0:
                # the per-step series: loss + macro metrics of THIS batch (mnist/models/mlp.py:33-106, every 20 steps)
                row = {f"{phase}/Loss": float(loss.detach())}
                if idx is not None:
                    bcm = torch.zeros(C * C, dtype=torch.int64).index_add_(0, idx, torch.ones_like(idx)).view(C, C)
                    row.update({f"{phase}/{k}": v for k, v in metrics_from_confusion(bcm).items() if k != "MicroAccuracy"})
                self._log(row, step=self._local_step + steps)
        return total / max(n, 1), cm, steps

    # ------------------------------------------------------------------ fit / evaluate
    def fit(self):
        try:
            if self.epochs <= 0:
                return
            self.create_trainer()
            self._local_step = 0
            if self._use_engine():
                eng = self._get_engine()
                self._model_to_engine(eng)
                out = eng.fit(self.epochs)
                self._engine_to_model(eng)
                eng.synchronize()
                self._local_step = eng.steps_per_epoch() * self.epochs
                per_step = eng.step_losses()
                clsf = getattr(self.model, "is_classifier", True)
                if per_step is not None:            # the per-step series, every 20 steps (lightninglearner.py:188;
                    per_cm = eng.step_metrics()     # mnist/models/mlp.py:33-106: loss + macro metrics of the batch)
                    for t in range(19, per_step.shape[0], 20):
                        row = {"Train/Loss": float(per_step[t, 0])}
                        row.update({f"Train/{k}": v for k, v in per_cm.get(t + 1, [{}])[0].items()})
                        self._log(row, step=t + 1)
                self._log_phase("Train", out["Train"]["loss"][0], out["Train"]["confusion"][0] if clsf else None,
                                self._local_step)
                if "Validation" in out:
                    self._log_phase("Validation", out["Validation"]["loss"][0],
                                    out["Validation"]["confusion"][0] if clsf else None, self._local_step)
                return
            self.model.to(self.device)
            self._optimizer = self.model.configure_optimizers()
            for _ in range(self.epochs):
                if self._stop.is_set():
                    break
                loss, cm, steps = self._run_loader(self.data.train_dataloader(), "Train", True)
                self._local_step += steps
                self._log_phase("Train", loss, cm, self._local_step)
                vloss, vcm, _ = self._run_loader(self.data.val_dataloader(), "Validation", False)
                if self.data.val_dataloader() is not None:
                    self._log_phase("Validation", vloss, vcm, self._local_step)
        except Exception as e:
            logging.error("Something went wrong with pytorch lightning. %s", e)
            logging.exception(e)

    def evaluate(self):
        try:
            if self.epochs <= 0:
                return None
            if self._use_engine():
                eng = self._get_engine()
                self._model_to_engine(eng)
                out = eng.evaluate()
                eng.synchronize()
                loss, cm = float(out["loss"][0]), (out["confusion"][0] if getattr(self.model, "is_classifier", True) else None)
            else:
                self.model.to(self.device)
                loss, cm, _ = self._run_loader(self.data.test_dataloader(), "Test", False)
            m = self._log_phase("Test", loss, cm, None)
            return loss, m.get("TestEpoch/Accuracy", 0.0)
        except Exception as e:
            logging.error("Something went wrong with pytorch lightning. %s", e)
            logging.exception(e)
            return None

    def log_validation_metrics(self, loss, metric, round=None, name=None):
        self._log({"Test/Loss": loss, "Test/Accuracy": metric}, step=round)
