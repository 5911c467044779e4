"""Weights & Biases logger (reference: fedstellar/learning/pytorch/remotelogger.py:31-344).

Same FL-aware stepping as the local loggers (remotelogger.py:105-106,163-174) and the same surface as the reference's
Lightning ``WandbLogger`` subclass: ``experiment`` (lazy ``wandb.init``, re-using a run that already exists,
:122-150), ``watch``, ``log_hyperparams`` / ``log_metrics`` (with ``trainer/global_step``, :163-174), ``log_table`` /
``log_text`` / ``log_image`` (:177-223), ``save_dir`` / ``name`` / ``version`` (:226-251), ``after_save_checkpoint`` +
checkpoint artifacts with aliases (:254-261, :310-344), ``use_artifact`` / ``download_artifact`` (:263-300) and
``finalize`` (:302-308).

``wandb`` is an optional dependency. Without it (this image has none, and no network) the logger keeps the run on
disk in a layout modelled on wandb's offline run directory, so a scenario configured with ``enable_remote_tracking``
still runs and nothing that was logged is lost::

    <save_dir>/wandb/offline-run-<YYYYmmdd_HHMMSS>-<id>/
        files/config.yaml            hyper-parameters (log_hyperparams)
        files/wandb-summary.json     last value of every metric
        files/wandb-metadata.json    project / group / name / start time / host
        files/media/table/<key>_<step>.table.json, files/media/images/<key>_<step>_<i>.(png|npy)
        history.jsonl                one JSON object per log call ({"_step", "_timestamp", metrics...})
        artifacts/<name>/<version>/  files of a logged artifact + manifest.json (sha256 per file, aliases, metadata)

("parity unpinned": the real offline format is wandb's binary ``.wandb`` datastore, which cannot be produced or
checked here; the wandb-present path is tested against a stand-in module in tests/test_loggers.py.)
"""
from __future__ import annotations

import hashlib
import json
import logging
import os
import shutil
import socket
import time
import uuid
from typing import Any, Dict, List, Optional

from .statisticslogger import _FederatedStepping


def _jsonable(v):
    try:
        import torch
        if isinstance(v, torch.Tensor):
            return v.item() if v.numel() == 1 else v.tolist()
    except ImportError:          # pragma: no cover
        pass
    if hasattr(v, "item") and not isinstance(v, (str, bytes)):
        try:
            return v.item()
        except (ValueError, TypeError):
            pass
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    return str(v)


class OfflineRun:
    """The on-disk run used when ``wandb`` is not importable (layout in the module docstring). It offers the part of
    ``wandb.sdk.wandb_run.Run`` that the logger and the node code touch: ``log``, ``config.update``, ``summary``,
    ``log_artifact`` / ``use_artifact``, ``finish``, ``id`` / ``name`` / ``project_name()``."""

    class _Config(dict):
        def update(self, d=None, allow_val_change=True, **kw):     # noqa: D401 - wandb's signature
            dict.update(self, _jsonable(dict(d or {}, **kw)))
            self._flush()

    def __init__(self, root: str, project: str, group: Optional[str], name: Optional[str], run_id: Optional[str] = None):
        self.id = run_id or uuid.uuid4().hex[:8]
        self.name, self.group, self._project = name or self.id, group, project
        self.dir = os.path.join(root, "wandb", f"offline-run-{time.strftime('%Y%m%d_%H%M%S')}-{self.id}")
        self.files = os.path.join(self.dir, "files")
        os.makedirs(os.path.join(self.files, "media"), exist_ok=True)
        self.summary: Dict[str, Any] = {}
        self.config = OfflineRun._Config()
        self.config._flush = self._write_config
        self._history = open(os.path.join(self.dir, "history.jsonl"), "a")
        self._step = 0
        self.finished = False
        with open(os.path.join(self.files, "wandb-metadata.json"), "w") as f:
            json.dump({"project": project, "group": group, "name": self.name, "id": self.id, "host": socket.gethostname(),
                       "startedAt": time.strftime("%Y-%m-%dT%H:%M:%S"), "mode": "offline (wandb not installed)"}, f, indent=1)

    def project_name(self) -> str:
        return self._project

    def _write_config(self):
        import yaml
        with open(os.path.join(self.files, "config.yaml"), "w") as f:
            yaml.safe_dump({k: {"value": v} for k, v in self.config.items()}, f, sort_keys=True)

    def log(self, data: Dict[str, Any], step: Optional[int] = None, commit: bool = True):
        if step is not None:
            self._step = int(step)
        row = {"_step": self._step, "_timestamp": time.time()}
        for k, v in data.items():
            row[k] = v = _jsonable(v)
            if isinstance(v, (int, float, str, bool)):
                self.summary[k] = v
        self._history.write(json.dumps(row) + "\n")
        self._history.flush()
        with open(os.path.join(self.files, "wandb-summary.json"), "w") as f:
            json.dump(self.summary, f, indent=1)
        if step is None:
            self._step += 1

    # ---- media ----------------------------------------------------------------------------------------------
    def save_table(self, key: str, columns, data, step: int) -> str:
        d = os.path.join(self.files, "media", "table")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"{key.replace('/', '_')}_{step}.table.json")
        with open(path, "w") as f:
            json.dump({"columns": list(columns or []), "data": _jsonable(list(data or []))}, f)
        return os.path.relpath(path, self.files)

    def save_image(self, key: str, img, step: int, i: int) -> str:
        d = os.path.join(self.files, "media", "images")
        os.makedirs(d, exist_ok=True)
        stem = os.path.join(d, f"{key.replace('/', '_')}_{step}_{i}")
        if isinstance(img, (str, os.PathLike)) and os.path.exists(img):
            path = stem + os.path.splitext(str(img))[1]
            shutil.copyfile(img, path)
        else:
            import numpy as np
            arr = img.detach().cpu().numpy() if hasattr(img, "detach") else np.asarray(img)
            path = stem + ".npy"
            np.save(path, arr, allow_pickle=False)
        return os.path.relpath(path, self.files)

    # ---- artifacts ------------------------------------------------------------------------------------------
    def _artifact_root(self, name: str) -> str:
        return os.path.join(self.dir, "artifacts", name.replace("/", "_"))

    def log_artifact(self, artifact: "OfflineArtifact", aliases: Optional[List[str]] = None) -> "OfflineArtifact":
        root = self._artifact_root(artifact.name)
        os.makedirs(root, exist_ok=True)
        version = f"v{len([d for d in os.listdir(root) if d.startswith('v')])}"
        dst = os.path.join(root, version)
        os.makedirs(dst)
        manifest = {"name": artifact.name, "type": artifact.type, "version": version, "metadata": _jsonable(artifact.metadata),
                    "aliases": sorted(set((aliases or []) + ["latest"])), "files": {}}
        for arc, src in artifact.entries:
            out = os.path.join(dst, arc)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            shutil.copyfile(src, out)
            with open(out, "rb") as f:
                manifest["files"][arc] = hashlib.sha256(f.read()).hexdigest()
        with open(os.path.join(dst, "manifest.json"), "w") as f:
            json.dump(manifest, f, indent=1)
        artifact.version, artifact.dir = version, dst
        return artifact

    def use_artifact(self, name: str, type: Optional[str] = None) -> "OfflineArtifact":
        """``name`` or ``name:alias`` / ``name:vN`` of an artifact logged to THIS run directory."""
        base, _, want = name.partition(":")
        root = self._artifact_root(base)
        versions = sorted((d for d in os.listdir(root) if d.startswith("v")), key=lambda d: int(d[1:])) \
            if os.path.isdir(root) else []
        for v in reversed(versions):
            with open(os.path.join(root, v, "manifest.json")) as f:
                m = json.load(f)
            if not want or want == v or want in m["aliases"]:
                a = OfflineArtifact(base, m["type"], m["metadata"])
                a.version, a.dir = v, os.path.join(root, v)
                return a
        raise FileNotFoundError(f"artifact {name!r} was not logged to {self.dir} (offline run: no server to ask)")

    def finish(self):
        if not self.finished:
            self.finished = True
            self._history.close()


class OfflineArtifact:
    def __init__(self, name: str, type: str = "model", metadata: Optional[dict] = None):
        self.name, self.type, self.metadata = name, type, dict(metadata or {})
        self.entries: List[tuple] = []
        self.version: Optional[str] = None
        self.dir: Optional[str] = None

    def add_file(self, path: str, name: Optional[str] = None):
        self.entries.append((name or os.path.basename(path), path))

    def download(self, root: Optional[str] = None) -> str:
        if self.dir is None:
            raise RuntimeError("artifact has not been logged")
        if root is None:
            return self.dir
        shutil.copytree(self.dir, root, dirs_exist_ok=True)
        return root


class FedstellarWBLogger(_FederatedStepping):
    def __init__(self, name=None, save_dir=".", version=None, offline=False, project="fedstellar", group=None,
                 log_model=False, prefix="", checkpoint_name=None, id=None, **kwargs):
        super().__init__()
        if offline and log_model == "all":
            raise ValueError("log_model='all' needs an online run (reference remotelogger.py:58-64)")
        self._name, self._project, self._group, self._save_dir = name, project, group, save_dir
        self._log_model, self._prefix, self._checkpoint_name = log_model, prefix, checkpoint_name
        self._logged_model_time: Dict[str, float] = {}
        self._checkpoint_callback = None
        self._run = None
        try:
            import wandb
            self._wandb = wandb
        except ImportError:
            self._wandb = None
            logging.warning("[W&B] wandb is not installed: keeping the run on disk under %s/wandb", save_dir)
        self._init_kwargs = dict(name=name, project=project, group=group, dir=save_dir, id=version or id,
                                 resume="allow", mode="offline" if offline else None, **kwargs)

    def __getstate__(self):
        """Picklable: the run handle (and the imported module) stay behind; the run's id travels so the copy re-attaches
        to the same run on first use (reference remotelogger.py:108-118)."""
        state = self.__dict__.copy()
        if self._run is not None:
            state["_init_kwargs"] = dict(self._init_kwargs, id=getattr(self._run, "id", None) or self._init_kwargs.get("id"))
        state["_run"] = None
        state["_wandb"] = None
        state["_checkpoint_callback"] = None
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        try:
            import wandb
            self._wandb = wandb
        except ImportError:
            self._wandb = None

    # ------------------------------------------------------------------ run handle
    @property
    def experiment(self):
        if self._run is None:
            if self._wandb is None:
                self._run = OfflineRun(self._save_dir, self._project, self._group, self._name, self._init_kwargs["id"])
            elif getattr(self._wandb, "run", None) is not None:
                logging.warning("[W&B] There is a wandb run already in progress: this logger re-uses it")
                self._run = self._wandb.run
            else:
                self._run = self._wandb.init(**{k: v for k, v in self._init_kwargs.items() if v is not None})
                if hasattr(self._run, "define_metric"):       # x axis of every chart (remotelogger.py:144-148)
                    self._run.define_metric("trainer/global_step")
                    self._run.define_metric("*", step_metric="trainer/global_step", step_sync=True)
        return self._run

    @property
    def save_dir(self) -> Optional[str]:
        return self._save_dir

    @property
    def name(self) -> Optional[str]:
        return self._run.project_name() if self._run is not None else self._project

    @property
    def version(self) -> Optional[str]:
        return self._run.id if self._run is not None else self._init_kwargs["id"]

    def _prefixed(self, metrics):
        return {f"{self._prefix}-{k}" if self._prefix else k: v for k, v in metrics.items()}

    # ------------------------------------------------------------------ logging
    def watch(self, model, log="gradients", log_freq=100, log_graph=True):
        if self._wandb is not None:
            self.experiment.watch(model, log=log, log_freq=log_freq, log_graph=log_graph)

    def log_metrics(self, metrics, step=None):
        s = self._step(step)
        self.experiment.log(dict(self._prefixed(metrics), **{"trainer/global_step": s}))

    def log_metrics_direct(self, metrics, step):
        self.log_metrics(metrics, step)

    def log_hyperparams(self, params, metrics=None):
        params = vars(params) if hasattr(params, "__dict__") and not isinstance(params, dict) else dict(params)
        self.experiment.config.update(_jsonable(params), allow_val_change=True)

    def log_table(self, key, columns=None, data=None, dataframe=None, step=None):
        if self._wandb is not None:
            self.log_metrics({key: self._wandb.Table(columns=columns, data=data, dataframe=dataframe)}, step)
            return
        if dataframe is not None:
            columns, data = list(dataframe.columns), dataframe.values.tolist()
        s = self._step(step)
        run = self.experiment
        run.log({key: {"_type": "table-file", "path": run.save_table(key, columns, data, s),
                       "nrows": len(data or []), "ncols": len(columns or [])}, "trainer/global_step": s})

    def log_text(self, key, columns=None, data=None, dataframe=None, step=None):
        self.log_table(key, columns, data, dataframe, step)

    def log_image(self, key, images, step=None, **kwargs):
        if not isinstance(images, list):
            raise TypeError(f'Expected a list as "images", found {type(images)}')
        n = len(images)
        for k, v in kwargs.items():
            if len(v) != n:
                raise ValueError(f"Expected {n} items but only found {len(v)} for {k}")
        if self._wandb is not None:
            per = [{k: kwargs[k][i] for k in kwargs} for i in range(n)]
            self.log_metrics({key: [self._wandb.Image(img, **kw) for img, kw in zip(images, per)]}, step)
            return
        s = self._step(step)
        run = self.experiment
        paths = [run.save_image(key, img, s, i) for i, img in enumerate(images)]
        run.log({key: {"_type": "images", "count": n, "paths": paths, "captions": _jsonable(kwargs.get("caption"))},
                 "trainer/global_step": s})

    def log_graph(self, model, input_array=None):
        pass

    # ------------------------------------------------------------------ checkpoints as artifacts
    def after_save_checkpoint(self, checkpoint_callback) -> None:
        """``checkpoint_callback``: anything with ``best_k_models`` {path: score} / ``best_model_path`` /
        ``last_model_path`` / ``save_top_k`` / ``monitor`` (Lightning's ModelCheckpoint, or the node's own record)."""
        if self._log_model == "all" or (self._log_model is True and getattr(checkpoint_callback, "save_top_k", 1) == -1):
            self._scan_and_log_checkpoints(checkpoint_callback)
        elif self._log_model is True:
            self._checkpoint_callback = checkpoint_callback

    def _scan_and_log_checkpoints(self, cb) -> None:
        found = dict(getattr(cb, "best_k_models", {}) or {})
        for attr in ("best_model_path", "last_model_path"):
            p = getattr(cb, attr, None)
            if p:
                found.setdefault(p, getattr(cb, "current_score", None))
        todo = sorted((os.path.getmtime(p), p, s) for p, s in found.items()
                      if os.path.isfile(p) and self._logged_model_time.get(p, -1.0) < os.path.getmtime(p))
        run = self.experiment
        for t, p, score in todo:
            meta = {"score": _jsonable(score), "original_filename": os.path.basename(p),
                    "monitor": getattr(cb, "monitor", None), "save_top_k": getattr(cb, "save_top_k", None)}
            name = self._checkpoint_name or f"model-{run.id}"
            art = (self._wandb.Artifact(name=name, type="model", metadata=meta) if self._wandb is not None
                   else OfflineArtifact(name, "model", meta))
            art.add_file(p, name="model.ckpt")
            aliases = ["latest", "best"] if p == getattr(cb, "best_model_path", None) else ["latest"]
            run.log_artifact(art, aliases=aliases)
            self._logged_model_time[p] = t

    def use_artifact(self, artifact: str, artifact_type: Optional[str] = None):
        return self.experiment.use_artifact(artifact, type=artifact_type)

    def download_artifact(self, artifact: str, save_dir: Optional[str] = None, artifact_type: Optional[str] = None,
                          use_artifact: bool = True) -> str:
        if self._wandb is not None and not use_artifact:
            art = self._wandb.Api().artifact(artifact, type=artifact_type)
        else:
            art = self.use_artifact(artifact, artifact_type)
        return art.download(root=save_dir)

    # ------------------------------------------------------------------ end of run
    def save(self):
        pass

    def finalize(self, status="success"):
        if status != "success":
            return                       # the reference leaves a failed run open for inspection (remotelogger.py:302-304)
        if self._checkpoint_callback is not None and self._run is not None:
            self._scan_and_log_checkpoints(self._checkpoint_callback)
        self.close()

    def close(self):
        if self._run is not None:
            self._run.finish()
