"""Weights & Biases logger (learning/pytorch/remotelogger.py): the on-disk run kept when ``wandb`` is absent, and the
wandb-present code path against a stand-in module (reference: fedstellar/learning/pytorch/remotelogger.py:31-344;
the real service / binary offline format cannot be reached or produced here: "parity unpinned")."""
import json
import sys
import types

import numpy as np
import pytest
import torch


def _fresh_logger_module(monkeypatch, wandb_module=None):
    import importlib
    if wandb_module is None:
        monkeypatch.setitem(sys.modules, "wandb", None)          # import wandb -> ImportError
    else:
        monkeypatch.setitem(sys.modules, "wandb", wandb_module)
    import p2pfl_amd.learning.pytorch.remotelogger as m
    return importlib.reload(m)


def test_offline_run_keeps_everything_on_disk(tmp_path, monkeypatch):
    m = _fresh_logger_module(monkeypatch)
    lg = m.FedstellarWBLogger(project="proj", group="scen", name="participant_3", save_dir=str(tmp_path), log_model=True)
    assert lg.save_dir == str(tmp_path) and lg.name == "proj"
    lg.log_hyperparams({"epochs": 3, "lr": torch.tensor(1e-3), "nested": {"a": (1, 2)}})
    lg.log_metrics({"Train/Loss": torch.tensor(0.5), "Train/Accuracy": np.float32(0.25)}, step=0)
    lg.log_metrics({"Train/Loss": 0.25}, step=1)
    lg.finalize_round()                                   # FL-aware stepping: the next round continues after step 1
    lg.log_metrics({"Train/Loss": 0.125}, step=0)
    lg.log_table("preds", columns=["x", "y"], data=[[1, "a"], [2, "b"]], step=0)
    lg.log_text("notes", columns=["t"], data=[["hello"]])
    lg.log_image("samples", [torch.zeros(1, 4, 4), np.ones((4, 4))], step=0, caption=["zero", "one"])
    with pytest.raises(ValueError):
        lg.log_image("samples", [np.ones((2, 2))], caption=["a", "b"])
    run = lg.experiment
    assert lg.version == run.id and run.project_name() == "proj"
    rows = [json.loads(l) for l in open(run.dir + "/history.jsonl")]
    steps = [r["trainer/global_step"] for r in rows if "Train/Loss" in r]
    assert steps == [0, 1, 1]                             # global_step += local_step at the end of a round (remotelogger.py:105-106)
    assert rows[0]["Train/Accuracy"] == 0.25
    summary = json.load(open(run.files + "/wandb-summary.json"))
    assert summary["Train/Loss"] == 0.125
    import yaml
    cfg = yaml.safe_load(open(run.files + "/config.yaml"))
    assert cfg["epochs"]["value"] == 3 and abs(cfg["lr"]["value"] - 1e-3) < 1e-9 and cfg["nested"]["value"] == {"a": [1, 2]}
    meta = json.load(open(run.files + "/wandb-metadata.json"))
    assert meta["group"] == "scen" and meta["name"] == "participant_3"
    table = [r for r in rows if "preds" in r][0]["preds"]
    assert table["nrows"] == 2 and json.load(open(run.files + "/" + table["path"]))["data"][1] == [2, "b"]
    imgs = [r for r in rows if "samples" in r][0]["samples"]
    assert imgs["count"] == 2 and np.load(run.files + "/" + imgs["paths"][1]).sum() == 16

    # checkpoints become versioned artifacts with aliases when the run ends (log_model=True)
    ck1, ck2 = tmp_path / "a.ckpt", tmp_path / "b.ckpt"
    ck1.write_bytes(b"one")
    ck2.write_bytes(b"two!")
    cb = types.SimpleNamespace(best_k_models={str(ck1): 0.9, str(ck2): 0.7}, best_model_path=str(ck2),
                               last_model_path=str(ck1), save_top_k=2, monitor="Validation/Loss", current_score=0.7)
    lg.after_save_checkpoint(cb)
    lg.finalize("failed")                                 # only successful runs log their checkpoints
    assert not list((tmp_path / "wandb").rglob("manifest.json"))
    lg.finalize("success")
    manifests = sorted((tmp_path / "wandb").rglob("manifest.json"))
    assert len(manifests) == 2 and run.finished
    best = [json.load(open(p)) for p in manifests if "best" in json.load(open(p))["aliases"]]
    assert len(best) == 1 and best[0]["metadata"]["original_filename"] == "b.ckpt" and best[0]["metadata"]["score"] == 0.7
    art = run.use_artifact(f"model-{run.id}:best")
    assert open(art.download() + "/model.ckpt", "rb").read() == b"two!"
    out = lg.download_artifact(f"model-{run.id}:v0", save_dir=str(tmp_path / "dl"))
    assert (tmp_path / "dl" / "model.ckpt").exists() and out == str(tmp_path / "dl")
    with pytest.raises(FileNotFoundError):
        run.use_artifact("nope")
    with pytest.raises(ValueError):
        m.FedstellarWBLogger(offline=True, log_model="all", save_dir=str(tmp_path))


def test_wandb_present_path_calls_the_sdk(tmp_path, monkeypatch):
    """With a ``wandb`` module importable the logger drives it the way the reference does: lazy init with the
    project / group / name / dir, a step metric, ``trainer/global_step`` in every log call, Table / Image wrappers,
    model artifacts with aliases, finish at the end."""
    calls = []

    class Run:
        id = "abc123"
        config = types.SimpleNamespace(update=lambda d, allow_val_change=False: calls.append(("config", d, allow_val_change)))

        def project_name(self):
            return "proj"

        def define_metric(self, *a, **k):
            calls.append(("define_metric", a, k))

        def log(self, d):
            calls.append(("log", d))

        def watch(self, model, **k):
            calls.append(("watch", k))

        def log_artifact(self, art, aliases=None):
            calls.append(("artifact", art.name, art.files, aliases))

        def use_artifact(self, name, type=None):
            calls.append(("use", name, type))
            return types.SimpleNamespace(download=lambda root=None: root or "/dl")

        def finish(self):
            calls.append(("finish",))

    class Artifact:
        def __init__(self, name, type, metadata):
            self.name, self.type, self.metadata, self.files = name, type, metadata, []

        def add_file(self, path, name=None):
            self.files.append(name)

    fake = types.ModuleType("wandb")
    fake.run = None
    fake.init = lambda **kw: calls.append(("init", kw)) or Run()
    fake.Table = lambda columns=None, data=None, dataframe=None: ("table", columns, data)
    fake.Image = lambda img, **kw: ("image", kw)
    fake.Artifact = Artifact
    m = _fresh_logger_module(monkeypatch, fake)
    lg = m.FedstellarWBLogger(project="proj", group="scen", name="participant_0", save_dir=str(tmp_path), log_model="all")
    assert calls == []                                    # nothing happens before the first use
    lg.log_metrics({"Train/Loss": 0.5}, step=4)
    assert calls[0][0] == "init" and calls[0][1]["project"] == "proj" and calls[0][1]["group"] == "scen" \
        and calls[0][1]["name"] == "participant_0" and calls[0][1]["dir"] == str(tmp_path) and "mode" not in calls[0][1]
    assert calls[1][0] == "define_metric" and calls[-1] == ("log", {"Train/Loss": 0.5, "trainer/global_step": 4})
    lg.log_hyperparams({"epochs": 3})
    assert calls[-1] == ("config", {"epochs": 3}, True)
    lg.log_table("t", columns=["a"], data=[[1]], step=5)
    assert calls[-1][1]["t"] == ("table", ["a"], [[1]])
    lg.log_image("i", [0, 1], step=6, caption=["x", "y"])
    assert calls[-1][1]["i"] == [("image", {"caption": "x"}), ("image", {"caption": "y"})]
    lg.watch(torch.nn.Linear(2, 2), log_freq=10)
    assert calls[-1] == ("watch", {"log": "gradients", "log_freq": 10, "log_graph": True})
    ck = tmp_path / "m.ckpt"
    ck.write_bytes(b"x")
    cb = types.SimpleNamespace(best_k_models={str(ck): 0.1}, best_model_path=str(ck), last_model_path="", save_top_k=1,
                               monitor="loss", current_score=0.1)
    lg.after_save_checkpoint(cb)                          # log_model="all": logged immediately
    assert calls[-1] == ("artifact", "model-abc123", ["model.ckpt"], ["latest", "best"])
    lg.after_save_checkpoint(cb)                          # unchanged file: not logged twice
    assert [c[0] for c in calls].count("artifact") == 1
    assert lg.download_artifact("model-abc123:best", save_dir="/x") == "/x" and calls[-1] == ("use", "model-abc123:best", None)
    assert lg.version == "abc123" and lg.name == "proj"
    lg.finalize("success")
    assert calls[-1] == ("finish",)
    # an already running wandb run is re-used instead of a second init (remotelogger.py:128-134)
    fake.run = Run()
    n_init = [c[0] for c in calls].count("init")
    lg2 = m.FedstellarWBLogger(project="proj", save_dir=str(tmp_path))
    assert lg2.experiment is fake.run and [c[0] for c in calls].count("init") == n_init


def test_tensorboard_logger_directory_layout_versions_and_pickling(tmp_path):
    """Layout and auto-versioning of the reference logger (statisticslogger.py:63-88,207-233): <save_dir>/<name>/
    version_<n>[/<sub_dir>], the next free n when no version is given; the logger pickles without its open file."""
    import pickle

    import numpy as np
    from p2pfl_amd.learning.pytorch.statisticslogger import FedstellarLogger, read_event_file
    a = FedstellarLogger(str(tmp_path), name="run")
    assert a.root_dir == str(tmp_path / "run") and a.version == 0 and a.log_dir == str(tmp_path / "run" / "version_0")
    a.log_metrics({"Test/Loss": 1.0}, step=1)
    a.finalize("success")
    b = FedstellarLogger(str(tmp_path), name="run", sub_dir="fold1")
    assert b.version == 1 and b.log_dir == str(tmp_path / "run" / "version_1" / "fold1") and b.sub_dir == "fold1"
    b.log_hyperparams({"lr": np.float32(0.5), "w": np.zeros((2, 2)), "steps": np.arange(3)})
    assert b.hparams["lr"] == 0.5 and isinstance(b.hparams["w"], str) and b.hparams["steps"] == [0, 1, 2]
    b.log_metrics({"Test/Loss": 2.0}, step=1)
    c = pickle.loads(pickle.dumps(b))                     # the open event file is not part of the state
    assert c._writer is None and c.log_dir == b.log_dir and c.global_step == b.global_step
    b.finalize("success")
    files = [f for f in (tmp_path / "run" / "version_1" / "fold1").iterdir() if "tfevents" in f.name]
    assert len(files) == 1 and read_event_file(str(files[0]))["Test/Loss"][0][1] == 2.0
    named = FedstellarLogger(str(tmp_path), name="metrics", version="participant_3")
    assert named.log_dir == str(tmp_path / "metrics" / "participant_3")


def test_wb_logger_pickles_and_reattaches_to_its_run(tmp_path):
    """reference remotelogger.py:108-118: the logger pickles without its run handle and keeps the run id."""
    import pickle
    from p2pfl_amd.learning.pytorch.remotelogger import FedstellarWBLogger
    lg = FedstellarWBLogger(name="n0", save_dir=str(tmp_path), project="proj", group="g", offline=True)
    lg.log_metrics({"Test/Loss": 0.5}, step=1)
    rid = lg.version
    copy = pickle.loads(pickle.dumps(lg))
    assert copy._run is None and copy.version == rid and copy.global_step == lg.global_step
    copy.log_metrics({"Test/Loss": 0.25}, step=2)        # first use re-opens the SAME run directory
    assert copy.version == rid
    lg.finalize("success"); copy.finalize("success")
