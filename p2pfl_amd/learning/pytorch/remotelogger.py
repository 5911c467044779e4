"""Weights & Biases logger (reference: fedstellar/learning/pytorch/remotelogger.py:31-344).

Same FL-aware stepping as the local loggers (remotelogger.py:105-106,163-174). ``wandb`` is an
optional dependency: without it (or without network) the logger degrades to the CSV logger under
``./wandb-offline`` so a scenario configured with ``enable_remote_tracking`` still runs.
"""
from __future__ import annotations

import logging
import os

from .statisticslogger import CSVLogger, _FederatedStepping


class FedstellarWBLogger(_FederatedStepping):
    def __init__(self, name=None, save_dir=".", version=None, offline=False, project="fedstellar", group=None,
                 log_model=False, **kwargs):
        super().__init__()
        self._name, self._project, self._group = name, project, group
        self._fallback = None
        self._run = None
        try:
            import wandb  # noqa: F401
            self._wandb = wandb
        except ImportError:
            self._wandb = None
            logging.warning("[W&B] wandb is not installed: logging to CSV under %s/wandb-offline", save_dir)
            self._fallback = CSVLogger(os.path.join(save_dir, "wandb-offline"), name=group or "run", version=name)
        self._init_kwargs = dict(name=name, project=project, group=group, dir=save_dir,
                                 mode="offline" if offline else None, **kwargs)

    @property
    def experiment(self):
        if self._wandb is None:
            return self._fallback
        if self._run is None:
            kw = {k: v for k, v in self._init_kwargs.items() if v is not None}
            self._run = self._wandb.init(**kw)
        return self._run

    def watch(self, model, log="gradients", log_freq=100, log_graph=True):
        if self._wandb is not None:
            self.experiment.watch(model, log=log, log_freq=log_freq, log_graph=log_graph)

    def log_metrics(self, metrics, step=None):
        s = self._step(step)
        if self._wandb is None:
            self._fallback.log_metrics_direct(metrics, s)
        else:
            self.experiment.log(dict(metrics, **{"trainer/global_step": s}))

    log_metrics_direct = lambda self, metrics, step: self.log_metrics(metrics, step)

    def log_hyperparams(self, params, metrics=None):
        if self._wandb is None:
            self._fallback.log_hyperparams(params)
        else:
            self.experiment.config.update(dict(params), allow_val_change=True)

    def log_table(self, key, columns=None, data=None, dataframe=None, step=None):
        if self._wandb is not None:
            self.log_metrics({key: self._wandb.Table(columns=columns, data=data, dataframe=dataframe)}, step)

    def log_text(self, key, columns=None, data=None, dataframe=None, step=None):
        self.log_table(key, columns, data, dataframe, step)

    def log_image(self, key, images, step=None, **kwargs):
        if self._wandb is not None:
            self.log_metrics({key: [self._wandb.Image(img) for img in images]}, step)

    def log_graph(self, model, input_array=None):
        pass

    def save(self):
        if self._fallback is not None:
            self._fallback.save()

    def finalize(self, status="success"):
        if self._fallback is not None:
            self._fallback.finalize(status)
        elif self._run is not None:
            self._run.finish()

    close = finalize
