"""Model families, data modules, learner and loggers (CPU).

The reference ships no tests; expectations come from its sources:
state_dict layouts / parameter counts (mnist/models/mlp.py:144-146, mnist/models/cnn.py:134-144,
femnist/models/cnn.py:136-140), contiguous sharding + 90/10 split (mnist/mnist.py:100-118), payload format and
learner contract (lightninglearner.py:73-165), FL-aware logger stepping (statisticslogger.py:60-61,132-136).
"""
import gzip
import os
import struct

import numpy as np
import pytest
import torch

from p2pfl_amd.node_start import build_dataset_and_model


def _reset_singletons():
    from p2pfl_amd.learning.pytorch.cifar10.cifar10 import CIFAR10DataModule
    from p2pfl_amd.learning.pytorch.femnist.femnist import FEMNISTDataModule
    from p2pfl_amd.learning.pytorch.mnist.mnist import MNISTDataModule
    from p2pfl_amd.learning.pytorch.syscall.syscall import SYSCALLDataModule
    from p2pfl_amd.learning.pytorch.wadi.wadi import WADIDataModule
    MNISTDataModule.mnist_train = MNISTDataModule.mnist_val = None
    FEMNISTDataModule.femnist_train = FEMNISTDataModule.femnist_val = None
    CIFAR10DataModule.cifar10_train = CIFAR10DataModule.cifar10_val = None
    SYSCALLDataModule.train_set = SYSCALLDataModule.test_set = None
    WADIDataModule.train_set = WADIDataModule.test_set = None


@pytest.fixture(autouse=True)
def fresh_datasets():
    _reset_singletons()
    yield
    _reset_singletons()


# ----------------------------------------------------------------------------------- models
LAYOUTS = {
    ("MNIST", "MLP"): (235146, ["l1.weight", "l1.bias", "l2.weight", "l2.bias", "l3.weight", "l3.bias"]),
    ("MNIST", "CNN"): (6497162, ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "l1.weight", "l1.bias",
                                 "l2.weight", "l2.bias"]),
    ("FEMNIST", "CNN"): (214590, ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "out.weight", "out.bias"]),
}


@pytest.mark.parametrize("ds,model", sorted(LAYOUTS))
def test_reference_state_dict_layouts(ds, model):
    data, m = build_dataset_and_model(ds, model, 0, 100)
    n_params, keys = LAYOUTS[(ds, model)]
    sd = m.state_dict()
    assert list(sd.keys()) == keys
    assert sum(v.numel() for v in sd.values()) == n_params
    # the batched engine's descriptor describes exactly this module
    spec = m.spec
    assert [p.name for p in spec.params] == keys and spec.num_params == n_params
    assert [tuple(p.shape) for p in spec.params] == [tuple(v.shape) for v in sd.values()]


FAMILIES = [("MNIST", "MLP", (1, 28, 28), 10), ("MNIST", "CNN", (1, 28, 28), 10), ("FEMNIST", "CNN", (1, 28, 28), 62),
            ("SYSCALL", "MLP", (17,), 9), ("SYSCALL", "SVM", (17,), None), ("SYSCALL", "Autoencoder", (17,), 17),
            ("CIFAR10", "ResNet9", (3, 32, 32), 10), ("CIFAR10", "ResNet18", (3, 32, 32), 10),
            ("CIFAR10", "fastermobilenet", (3, 32, 32), 10), ("CIFAR10", "simplemobilenet", (3, 32, 32), 10),
            ("WADI", "MLP", (123,), 1)]


@pytest.mark.parametrize("ds,model,shape,out", FAMILIES)
def test_every_family_forward_loss_backward(ds, model, shape, out):
    data, m = build_dataset_and_model(ds, model, 1, 200)
    assert data.input_shape == shape
    x = data.train_x[:8].reshape(8, *shape)
    y = data.train_y[:8]
    m.train()
    o = m(x)
    assert o.shape[0] == 8 and (out is None or o.shape[-1] == out)
    loss, pred = m.training_loss(x, y)
    assert torch.isfinite(loss)
    loss.backward()
    grads = [p.grad for p in m.parameters() if p.requires_grad]
    assert grads and all(g is not None and torch.isfinite(g).all() for g in grads)
    opt = m.configure_optimizers()
    opt.step()
    if m.is_classifier:
        assert pred is not None and pred.shape == (8,)


def test_reference_optimizers():
    """Adam 1e-3 (MNIST), SGD 1e-3 m=0.9 (FEMNIST), Adam 1e-2 (syscall MLP/AE), SGD 1e-2 (SVM),
    Adam 1e-2 wd 1e-4 (ResNet) - mlp.py:164-167, femnist cnn.py:153-157, syscall mlp.py:155-158, svm.py:155-157,
    resnet.py:236-239."""
    def opt(ds, model):
        return build_dataset_and_model(ds, model, 0, 500)[1].configure_optimizers()
    o = opt("MNIST", "CNN")
    assert isinstance(o, torch.optim.Adam) and o.defaults["lr"] == 1e-3
    o = opt("FEMNIST", "CNN")
    assert isinstance(o, torch.optim.SGD) and o.defaults["lr"] == 1e-3 and o.defaults["momentum"] == 0.9
    o = opt("SYSCALL", "MLP")
    assert isinstance(o, torch.optim.Adam) and o.defaults["lr"] == 1e-2
    o = opt("SYSCALL", "SVM")
    assert isinstance(o, torch.optim.SGD) and o.defaults["lr"] == 1e-2
    o = opt("CIFAR10", "ResNet9")
    assert isinstance(o, torch.optim.Adam) and o.defaults["lr"] == 1e-2 and o.defaults["weight_decay"] == 1e-4


def test_unknown_dataset_or_model_is_rejected():
    with pytest.raises(ValueError):
        build_dataset_and_model("IMAGENET", "MLP", 0, 2)
    with pytest.raises(ValueError):
        build_dataset_and_model("FEMNIST", "MLP", 0, 2)


def test_mnist_cnn_returns_logits_and_autoencoder_reconstructs():
    """The two reference forward() bugs (cnn.py:159, autoencoder.py:153-157) are not reproduced."""
    _, cnn = build_dataset_and_model("MNIST", "CNN", 0, 500)
    assert cnn(torch.rand(4, 1, 28, 28)).shape == (4, 10)
    _, ae = build_dataset_and_model("SYSCALL", "Autoencoder", 0, 500)
    x = torch.rand(4, 17)
    assert ae(x).shape == x.shape and not torch.equal(ae(x), x)
    assert ae.training_loss(x, None)[0] > 0


# ----------------------------------------------------------------------------------- data
def test_contiguous_sharding_split_sizes_and_non_iid():
    from p2pfl_amd.datasets.synthetic import reference_shard_sizes
    from p2pfl_amd.learning.pytorch.mnist.mnist import MNISTDataModule
    for n in (10, 100):
        d = MNISTDataModule(sub_id=3, number_sub=n)
        sizes = (len(d.train_x), len(d.val_x), len(d.test_x))
        assert sizes == reference_shard_sizes(n)
    assert reference_shard_sizes(10) == (5400, 600, 1000) and reference_shard_sizes(100) == (540, 60, 100)
    # shards of different nodes are disjoint blocks of the same singleton
    a, b = MNISTDataModule(0, 10), MNISTDataModule(1, 10)
    full = MNISTDataModule.mnist_train[0]
    assert torch.equal(torch.sort(torch.cat([a.train_x, a.val_x]).sum(1))[0], torch.sort(full[:6000].sum(1))[0])
    assert torch.equal(torch.sort(torch.cat([b.train_x, b.val_x]).sum(1))[0], torch.sort(full[6000:12000].sum(1))[0])
    # non-IID = label-sorted before cutting (mnist.py:76-83): a 10-way split sees ~1 class per node
    nid = MNISTDataModule(0, 10, iid=False)
    assert nid.train_y.unique().numel() <= 2 < a.train_y.unique().numel()
    with pytest.raises(ValueError):
        MNISTDataModule(sub_id=10, number_sub=10)
    # loaders: batch 32, image-shaped
    xb, yb = next(iter(a.train_dataloader()))
    assert xb.shape == (32, 1, 28, 28) and yb.shape == (32,)


def test_mnist_idx_files_are_read(tmp_path):
    from p2pfl_amd.learning.pytorch.mnist.mnist import MNISTDataModule
    raw = tmp_path / "MNIST" / "raw"
    os.makedirs(raw)
    rng = np.random.default_rng(0)

    def write(name, arr, gz):
        hdr = struct.pack(">I", 0x0800 | arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape)
        opener = gzip.open if gz else open
        with opener(raw / (name + (".gz" if gz else "")), "wb") as f:
            f.write(hdr + arr.tobytes())
    tx = rng.integers(0, 256, (200, 28, 28), dtype=np.uint8)
    ty = rng.integers(0, 10, 200, dtype=np.uint8)
    ex = rng.integers(0, 256, (40, 28, 28), dtype=np.uint8)
    ey = rng.integers(0, 10, 40, dtype=np.uint8)
    write("train-images-idx3-ubyte", tx, True)
    write("train-labels-idx1-ubyte", ty, False)
    write("t10k-images-idx3-ubyte", ex, False)
    write("t10k-labels-idx1-ubyte", ey, True)
    d = MNISTDataModule(1, 2, root_dir=str(tmp_path))
    assert not MNISTDataModule.synthetic
    assert len(d.train_x) + len(d.val_x) == 100 and len(d.test_x) == 20
    assert torch.equal(d.test_y, torch.from_numpy(ey[20:].astype(np.int64)))
    assert torch.allclose(d.test_x, torch.from_numpy(ex[20:].reshape(20, -1).astype(np.float32) / 255.0))


def test_syscall_raw_csv_processing(tmp_path):
    """Reference raw format: tab-separated, feature column is a list literal, `normalv2` folds into `normal`."""
    from p2pfl_amd.learning.pytorch.syscall.syscall import FEATURE, SYSCALLDataModule
    raw = tmp_path / "raw"
    os.makedirs(raw)
    rng = np.random.default_rng(1)
    kinds = ["normal", "normalv2", "bashlite", "thetick"]
    for fi in range(2):
        with open(raw / f"part{fi}.csv", "w") as f:
            f.write(f"timestamp\t{FEATURE}\tmaltype\n")
            for i in range(100):
                f.write(f"{i}\t{[round(float(v), 4) for v in rng.random(17)]}\t{kinds[(i + fi) % 4]}\n")
    d = SYSCALLDataModule(0, 2, root_dir=str(tmp_path))
    assert SYSCALLDataModule.classes == ["bashlite", "normal", "thetick"] and not SYSCALLDataModule.synthetic
    tr, te = SYSCALLDataModule.train_set, SYSCALLDataModule.test_set
    assert len(tr[0]) == 170 and len(te[0]) == 30 and tr[0].shape[1] == 17
    assert (tr[1] == 1).sum() + (te[1] == 1).sum() == 100            # normal + normalv2
    assert len(d.train_x) + len(d.val_x) == 85 and round(85 * 0.01) == len(d.val_x)


def test_femnist_transform_and_cifar_augment():
    from p2pfl_amd.learning.pytorch.cifar10.cifar10 import CIFAR10DataModule
    from p2pfl_amd.learning.pytorch.femnist.femnist import femnist_transform
    x = torch.rand(5, 28, 28)
    t = femnist_transform(x)
    assert t.shape == (5, 784)
    # CenterCrop(96) of a 28x28 image zero-pads it; Resize(28) shrinks the glyph to the centre ~8x8 pixels
    img = (t * 0.3081 + 0.1307).reshape(5, 28, 28)
    assert img[:, :8, :].abs().max() < 1e-4 and img[:, 11:17, 11:17].mean() > 0.2
    g = torch.Generator().manual_seed(0)
    b = torch.rand(6, 3, 32, 32)
    a = CIFAR10DataModule.augment(b, generator=g)
    assert a.shape == b.shape and not torch.equal(a, b)
    d = CIFAR10DataModule(0, 100)
    assert d.val_x is d.test_x                                   # reference: val loader == test loader
    assert len(d.train_dataloader()) == len(d.train_x) // 32      # drop_last


# ----------------------------------------------------------------------------------- learner
def _learner(ds="MNIST", model="MLP", logger=None, n=200):
    from p2pfl_amd.config.config import Config, default_participant
    from p2pfl_amd.learning.pytorch.lightninglearner import LightningLearner
    p = default_participant()
    p["device_args"]["accelerator"] = "cpu"
    data, m = build_dataset_and_model(ds, model, 0, n)
    return LightningLearner(m, data, config=Config(participant=p), logger=logger)


def test_learner_contract_fit_evaluate_payload():
    from p2pfl_amd.learning.exceptions import DecodingParamsError, ModelNotMatchingError
    ln = _learner()
    assert ln.get_num_samples() == (270, 50)
    ln.set_epochs(2)
    ln.init()
    ln.create_trainer()
    loss0, acc0 = ln.evaluate()
    ln.fit()
    loss1, acc1 = ln.evaluate()
    assert loss1 < loss0 and acc1 >= acc0
    # payload: pickle((arrays in state_dict order, contributors, weight)) and back
    blob = ln.encode_parameters(contributors=["a:1", "b:2"], weight=270)
    sd, contributors, weight = ln.decode_parameters(blob)
    assert contributors == ["a:1", "b:2"] and weight == 270 and ln.check_parameters(sd)
    assert all(torch.equal(sd[k], v) for k, v in ln.get_parameters().items())
    # a different architecture is refused
    other = _learner("MNIST", "CNN")
    sd2, _, _ = None, None, None
    with pytest.raises(DecodingParamsError):
        other.decode_parameters(blob)
    assert not other.check_parameters(sd)
    with pytest.raises(ModelNotMatchingError):
        other.set_parameters(sd)
    with pytest.raises(DecodingParamsError):
        ln.decode_parameters(b"not a pickle")
    # epochs == 0: nothing happens (reference lightninglearner.py:114,128)
    ln.set_epochs(0)
    before = {k: v.clone() for k, v in ln.get_parameters().items()}
    ln.fit()
    assert ln.evaluate() is None and all(torch.equal(before[k], v) for k, v in ln.get_parameters().items())


def test_payload_keeps_zero_dim_entries_of_batchnorm_models():
    """BatchNorm's ``num_batches_tracked`` is a 0-d tensor: it must come back 0-d or the receiver refuses the
    model (check_parameters compares shapes)."""
    ln = _learner("SYSCALL", "MLP")
    ln.set_epochs(1)
    ln.init()
    ln.fit()
    sd, _, _ = ln.decode_parameters(ln.encode_parameters(contributors=["a:1"], weight=3))
    assert sd["batchnorm1.num_batches_tracked"].shape == () and ln.check_parameters(sd)
    ln.set_parameters(sd)


def test_learner_interrupt_and_non_classifier_models():
    ln = _learner()
    ln.set_epochs(50)
    ln.interrupt_fit()          # flag is cleared by create_trainer inside fit; interrupt from another thread:
    import threading
    t = threading.Timer(0.3, ln.interrupt_fit)
    t.start()
    ln.fit()                    # returns long before 50 epochs
    t.join()
    for model in ("SVM", "Autoencoder"):
        l2 = _learner("SYSCALL", model, n=100)
        l2.set_epochs(1)
        l0 = l2.evaluate()[0]
        l2.fit()
        assert l2.evaluate()[0] <= l0 + 1e-6


def test_loggers_fl_stepping_and_files(tmp_path):
    from p2pfl_amd.learning.pytorch.statisticslogger import CSVLogger, FedstellarLogger
    lg = CSVLogger(str(tmp_path), name="metrics", version="participant_0")
    ln = _learner(logger=lg)
    ln.set_epochs(1)
    for _ in range(2):
        ln.evaluate()
        ln.fit()
        ln.finalize_round()
    lg.finalize("success")
    rows = (tmp_path / "metrics" / "participant_0" / "metrics.csv").read_text().strip().splitlines()
    assert rows[0] == "wall_time,step,key,value"
    names = {r.split(",")[2] for r in rows[1:]}
    assert {"Round", "TrainEpoch/Loss", "TrainEpoch/Accuracy", "ValidationEpoch/F1Score", "TestEpoch/Precision",
            "TestEpoch/Recall"} <= names
    # FL-aware stepping: steps of round 2 continue after those of round 1 (global_step + local_step)
    steps = [int(r.split(",")[1]) for r in rows[1:] if r.split(",")[2] == "TrainEpoch/Loss"]
    assert len(steps) == 2 and steps[1] == 2 * steps[0] and lg.global_step == steps[1]
    # the per-step series of the reference's models (torchmetrics of the step's batch every 20 steps,
    # mnist/models/mlp.py:33-106): loss AND macro metrics, at step 20 of a 34-step epoch
    lg2 = CSVLogger(str(tmp_path), name="metrics", version="participant_2")
    l2 = _learner(n=50, logger=lg2)
    l2.set_epochs(1)
    l2.fit()
    lg2.finalize("success")
    rows2 = [r.split(",") for r in (tmp_path / "metrics" / "participant_2" / "metrics.csv").read_text().strip().splitlines()[1:]]
    at20 = {r[2]: float(r[3]) for r in rows2 if int(r[1]) == 20}
    assert {"Train/Loss", "Train/Accuracy", "Train/Precision", "Train/Recall", "Train/F1Score"} <= set(at20)
    assert all(0.0 <= at20[k] <= 1.0 for k in at20 if k != "Train/Loss")
    # TensorBoard event file: TFRecord framing with masked crc32c, readable scalar events
    tb = FedstellarLogger(str(tmp_path), name="tb", version="participant_1")
    tb.log_metrics({"Test/Loss": 1.5}, step=3)
    tb.log_hyperparams({"lr": 1e-3})
    tb.finalize("success")
    files = list((tmp_path / "tb" / "participant_1").glob("events.out.tfevents.*"))
    assert len(files) == 1
    blob = files[0].read_bytes()
    n, = struct.unpack("<Q", blob[:8])
    assert 0 < n < 200 and b"Test/Loss" in blob and b"brain.Event:2" in blob


def test_wandb_logger_degrades_to_offline_files(tmp_path):
    """wandb is not installed on the target image: the remote logger must keep metrics locally."""
    from p2pfl_amd.learning.pytorch.remotelogger import FedstellarWBLogger
    lg = FedstellarWBLogger(project="p", group="g", name="participant_0", save_dir=str(tmp_path))
    lg.log_metrics({"Train/Loss": 0.5}, step=1)
    lg.log_hyperparams({"epochs": 3})
    lg.finalize_round()
    lg.finalize("success")
    assert any(tmp_path.rglob("*"))


@pytest.mark.parametrize("ds,model", [("MNIST", "MLP"), ("FEMNIST", "CNN"), ("SYSCALL", "Autoencoder"), ("CIFAR10", "simplemobilenet")])
def test_models_carry_the_references_step_and_metric_hooks(ds, model):
    """training_step / validation_step / test_step, process_metrics, log_metrics_by_epoch and the on_*_epoch_end hooks
    of the reference's LightningModules (mnist/models/mlp.py:33-222) on every model family; state_dict untouched."""
    data, m = build_dataset_and_model(ds, model, 0, 20)
    keys = list(m.state_dict())
    xb, yb = next(iter(data.train_dataloader()))
    rows = []

    class Lg:
        def log_metrics(self, metrics, step=None):
            rows.append(dict(metrics))
    m.logger = Lg()
    loss = m.training_step((xb, yb), 0)
    assert loss.requires_grad and float(loss.detach()) > 0
    assert m.logged_metrics["Train/Loss"] == pytest.approx(float(loss.detach()))
    with torch.no_grad():
        m.validation_step((xb, yb), 0)
        m.test_step((xb, yb), 0)
    if m.is_classifier:
        assert {"Train/Accuracy", "Train/Precision", "Train/Recall", "Train/F1Score", "Validation/F1Score",
                "Test/Accuracy"} <= set(m.logged_metrics)
        assert all(0.0 <= v <= 1.0 for k, v in m.logged_metrics.items() if not k.endswith("Loss"))
    m.on_train_epoch_end(); m.on_validation_epoch_end(); m.on_test_epoch_end()
    assert m.epoch_global_number == {"Train": 1, "Validation": 1, "Test": 1}
    if m.is_classifier:
        assert "TrainEpoch/Accuracy" in m.logged_metrics and "TestEpoch/F1Score" in m.logged_metrics
    assert rows and list(m.state_dict()) == keys and str(m.device) == "cpu"
    with pytest.raises(NotImplementedError):
        m.process_metrics("Holdout", None, yb)


def test_learner_drives_a_lightning_shaped_model_of_the_user():
    """A model written against the reference's Trainer (training_step / validation_step / test_step that return the
    loss and call self.log; no hooks of this package) trains and evaluates through LightningLearner."""
    from p2pfl_amd.config.config import Config, default_participant
    from p2pfl_amd.learning.pytorch.lightninglearner import LightningLearner

    class Mine(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = torch.nn.Linear(784, 10)

        def forward(self, x):
            return self.l(x.reshape(len(x), -1))

        def _step(self, batch, name):
            x, y = batch
            loss = torch.nn.functional.cross_entropy(self(x), y)
            self.log(f"{name}/MyLoss", loss)
            return loss

        def training_step(self, batch, i):
            return {"loss": self._step(batch, "Train")}

        def validation_step(self, batch, i):
            return self._step(batch, "Validation")

        def test_step(self, batch, i):
            return self._step(batch, "Test")

        def configure_optimizers(self):
            return torch.optim.SGD(self.parameters(), lr=0.1)

    rows = []

    class Lg:
        global_step = local_step = 0

        def log_metrics(self, metrics, step=None):
            rows.append(dict(metrics))
    p = default_participant()
    p["device_args"]["accelerator"] = "cpu"
    data, _ = build_dataset_and_model("MNIST", "MLP", 0, 200)
    ln = LightningLearner(Mine(), data, config=Config(participant=p), logger=Lg())
    ln.set_epochs(2)
    l0, _ = ln.evaluate()
    ln.fit()
    l1, _ = ln.evaluate()
    assert l1 < l0
    assert any("Train/MyLoss" in r for r in rows) and any("Test/MyLoss" in r for r in rows)


def test_dataset_objects_of_the_reference(tmp_path):
    """FEMNIST / SYSCALL / WADI dataset objects (femnist.py:24-82, syscall.py:21-120, wadi.py:20-60) on small local
    files: missing files raise unless download=True; FEMNIST is fetched from a file:// archive (checksum-verified,
    members confined to raw/), SYSCALL is processed from the raw tab-separated CSVs; sort_dataset orders by label."""
    import hashlib
    import tarfile

    import numpy as np
    from p2pfl_amd.learning.pytorch.femnist.femnist import FEMNIST
    from p2pfl_amd.learning.pytorch.syscall.syscall import FEATURE, SYSCALL, SYSCALLDataModule, sort_dataset
    from p2pfl_amd.learning.pytorch.wadi.wadi import WADI
    root = str(tmp_path)
    with pytest.raises(RuntimeError, match="download=True"):
        FEMNIST(root_dir=root)
    # FEMNIST: an archive with the two processed files, served from a file:// URL
    g = torch.Generator().manual_seed(0)
    src = tmp_path / "src"
    src.mkdir()
    for name, n in (("femnist_train.pt", 12), ("femnist_test.pt", 5)):
        torch.save((torch.rand(n, 28, 28, generator=g), torch.randint(0, 62, (n,), generator=g)), src / name)
    arc = tmp_path / "femnist.tar.gz"
    with tarfile.open(arc, "w:gz") as t:
        t.add(src / "femnist_train.pt", arcname="femnist_train.pt")
        t.add(src / "femnist_test.pt", arcname="femnist_test.pt")
        t.add(src / "femnist_test.pt", arcname="../escaped.pt")                # must not leave raw/
    md5 = hashlib.md5(arc.read_bytes()).hexdigest()

    class LocalFEMNIST(FEMNIST):
        download_link, file_md5 = "file://" + str(arc), md5
    tr = LocalFEMNIST(root_dir=root, download=True)
    te = LocalFEMNIST(root_dir=root, train=False, transform=lambda x: x.flatten(), target_transform=lambda y: y + 100)
    assert len(tr) == 12 and len(te) == 5 and tr[0][0].shape == (28, 28) and isinstance(tr[0][1], int)
    assert te[0][0].shape == (784,) and te[0][1] >= 100
    assert not (tmp_path / "FEMNIST" / "escaped.pt").exists() and not (tmp_path / "escaped.pt").exists()

    class Tampered(FEMNIST):
        download_link, file_md5 = "file://" + str(arc), "0" * 32
    with pytest.raises(RuntimeError, match="checksum"):
        Tampered(root_dir=str(tmp_path / "other"), download=True)
    # SYSCALL: raw CSVs -> processed files; the data module then reads what the dataset object wrote
    raw = tmp_path / "syscall" / "raw"
    raw.mkdir(parents=True)
    rows = ["id\t%s\tmaltype" % FEATURE]
    for i in range(40):
        rows.append("%d\t%s\t%s" % (i, [round(0.01 * ((i * 7 + j) % 13), 3) for j in range(17)],
                                     ["normal", "normalv2", "bashlite", "thetick"][i % 4]))
    (raw / "a.csv").write_text("\n".join(rows) + "\n")

    class LocalSYSCALL(SYSCALL):
        def dataset_download(self):          # the raw files are already in place
            os.makedirs(os.path.join(self.base, "processed"), exist_ok=True)
    s_tr, s_te = LocalSYSCALL(root_dir=root, download=True), LocalSYSCALL(root_dir=root, train=False)
    assert len(s_tr) == 34 and len(s_te) == 6 and s_tr.classes == ["bashlite", "normal", "thetick"]
    assert s_tr[0][0].shape == (17,) and s_tr.classes_to_targets["normal"] == 1
    sort_dataset(s_tr)
    assert bool((s_tr.targets[1:] >= s_tr.targets[:-1]).all())
    SYSCALLDataModule.train_set = None
    try:
        dm = SYSCALLDataModule(0, 2, root_dir=str(tmp_path / "syscall"))
        assert not SYSCALLDataModule.synthetic and dm.num_samples() == (17, 3)
    finally:
        SYSCALLDataModule.train_set = None
    # WADI: numpy files; no download location exists (the reference's is a placeholder)
    with pytest.raises(RuntimeError, match="no download location"):
        WADI(root_dir=root, download=True)
    (tmp_path / "WADI").mkdir()
    for name, shape in (("X_train.npy", (9, 123)), ("y_train.npy", (9,)), ("X_test.npy", (4, 123)), ("y_test.npy", (4,))):
        np.save(tmp_path / "WADI" / name, np.ones(shape))
    w = WADI(root_dir=root, train=False)
    assert len(w) == 4 and w[0][0].dtype == torch.float32 and w[0][1] == 1


def test_cifar10_normalization_and_whole_dataset_access():
    from p2pfl_amd.learning.pytorch.cifar10.cifar10 import CIFAR10DataModule
    assert CIFAR10DataModule.set_normalization("imagenet")["mean"] == (0.485, 0.456, 0.406)
    with pytest.raises(NotImplementedError):
        CIFAR10DataModule.set_normalization("other")
    with pytest.raises(NotImplementedError):
        CIFAR10DataModule(0, 50, loading="custom")
    a = CIFAR10DataModule(0, 50)
    b = CIFAR10DataModule(0, 50, normalization="imagenet")
    assert a.mean == (0.4914, 0.4822, 0.4465) and b.std == (0.229, 0.224, 0.225)
    ra = a.train_x.reshape(-1, 3, 32, 32) * torch.tensor(a.std).view(1, 3, 1, 1) + torch.tensor(a.mean).view(1, 3, 1, 1)
    rb = b.train_x.reshape(-1, 3, 32, 32) * torch.tensor(b.std).view(1, 3, 1, 1) + torch.tensor(b.mean).view(1, 3, 1, 1)
    assert torch.allclose(ra, rb, atol=1e-5)                      # the same pixels under both statistics
    whole = a.get_dataset(train=True)
    assert len(whole) == 50 * len(a.train_x) and whole[0][0].shape == (3, 32, 32)
    flipped = a.get_dataset(train=False, transform=lambda x: x.flip(-1))
    assert torch.equal(flipped[3][0], a.get_dataset(train=False)[3][0].flip(-1))


def test_wadi_mlp_has_a_kernel_level_spec():
    """The WADI MLP joins MNIST MLP / CNN and FEMNIST CNN on the client-batched kernels: its spec lists the module's
    state_dict in order, one output = two classes, and the CPU side of the spec (EagerEngine) trains it with the
    module's own sigmoid + BCE loss."""
    from p2pfl_amd.learning.pytorch.wadi.models.mlp import WADIModelMLP
    from p2pfl_amd.models.specs import get_spec
    from p2pfl_amd.models.torch_models import module_for_spec
    from p2pfl_amd.runtime.engine import EagerEngine
    from p2pfl_amd.runtime.launch import _GPU_ONLY_SPECS, _SPEC_OF
    spec, m = get_spec("wadi-mlp"), WADIModelMLP()
    sd = m.state_dict()
    assert [p.name for p in spec.params] == list(sd) and all(tuple(sd[p.name].shape) == p.shape for p in spec.params)
    assert m.spec.name == "wadi-mlp" and spec.num_classes == 2 and spec.layers[-1].out_features == 1
    assert [l.relu for l in spec.layers] == [True] * 8 + [False]
    assert isinstance(module_for_spec(spec), WADIModelMLP)
    assert _SPEC_OF[("WADI", "MLP")] == "wadi-mlp" and ("WADI", "MLP") in _GPU_ONLY_SPECS
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 123, generator=g)
    y = (x[:, :, 0] > 0).long()
    eng = EagerEngine(spec, 2, "cpu", batch_size=16, seed=1)
    eng.init_params()
    eng.set_data(x, y, x[:, :16], y[:, :16], x[:, :32], y[:, :32])
    l0 = eng.evaluate()["loss"].clone()
    out = eng.fit(3)
    assert out["Train"]["confusion"].shape == (2, 2, 2) and int(out["Train"]["count"][0]) == 64
    assert bool((eng.evaluate()["loss"] < l0).all())


def test_syscall_autoencoder_has_a_kernel_level_spec():
    """The auto-encoder's spec: state_dict order, MSE-against-the-input loss, and its CPU side (EagerEngine through the
    module's own training_loss: no predictions, so no class counters)."""
    from p2pfl_amd.learning.pytorch.syscall.models.autoencoder import SyscallModelAutoencoder
    from p2pfl_amd.models.specs import get_spec
    from p2pfl_amd.runtime.engine import EagerEngine
    from p2pfl_amd.runtime.launch import _GPU_ONLY_SPECS, _SPEC_OF
    spec, m = get_spec("syscall-autoencoder"), SyscallModelAutoencoder()
    sd = m.state_dict()
    assert [p.name for p in spec.params] == list(sd) and all(tuple(sd[p.name].shape) == p.shape for p in spec.params)
    assert spec.loss == "mse_input" and m.spec.name == "syscall-autoencoder" and get_spec("mnist-cnn").loss == "ce"
    assert [l.relu for l in spec.layers] == [True] * 5 + [False]
    assert _SPEC_OF[("SYSCALL", "Autoencoder")] == "syscall-autoencoder" and ("SYSCALL", "Autoencoder") in _GPU_ONLY_SPECS
    g = torch.Generator().manual_seed(0)
    x = torch.rand(2, 48, 17, generator=g)
    y = torch.randint(0, 9, (2, 48), generator=g)
    eng = EagerEngine(spec, 2, "cpu", batch_size=16, seed=1, log_steps=True)
    eng.init_params()
    eng.set_data(x, y, x[:, :16], y[:, :16], x[:, :32], y[:, :32])
    l0 = eng.evaluate()["loss"].clone()
    out = eng.fit(4)
    assert int(out["Train"]["confusion"].sum()) == 0 and int(out["Train"]["count"][0]) == 48
    assert bool((eng.evaluate()["loss"] < l0).all()) and eng.step_metrics() == {} and eng.step_losses().shape == (12, 2)


def test_archive_dataset_zip_members_stay_inside_raw_and_cm_tables_reach_the_logger(tmp_path):
    """dataset_download on a zip archive (the SYSCALL data ships as one): members that would leave raw/ are skipped;
    log_metrics_by_epoch(plot_cm=True) hands the epoch's confusion matrix to a logger that takes tables."""
    import zipfile
    from p2pfl_amd.learning.pytorch.syscall.syscall import FEATURE, SYSCALL
    arc = tmp_path / "syscall.zip"
    rows = ["id\t%s\tmaltype" % FEATURE] + ["%d\t%s\t%s" % (i, [0.1 * (i % 5)] * 17, ["normal", "bdvl"][i % 2]) for i in range(20)]
    with zipfile.ZipFile(arc, "w") as z:
        z.writestr("a.csv", "\n".join(rows) + "\n")
        z.writestr("../../evil.csv", "x")

    class Local(SYSCALL):
        download_link = "file://" + str(arc)
    ds = Local(root_dir=str(tmp_path / "data"), download=True)
    assert len(ds) == 17 and ds.classes == ["bdvl", "normal"]
    assert not (tmp_path / "evil.csv").exists() and not (tmp_path / "data" / "evil.csv").exists()
    # confusion-matrix table of an epoch
    _, m = build_dataset_and_model("MNIST", "MLP", 0, 50)
    tables = []

    class Lg:
        def log_metrics(self, metrics, step=None):
            pass

        def log_table(self, key, columns=None, data=None, step=None):
            tables.append((key, columns, data, step))
    m.logger = Lg()
    x, y = torch.randn(6, 1, 28, 28), torch.tensor([0, 1, 2, 3, 4, 5])
    m.training_step((x, y), 0)
    m.on_train_epoch_end()
    key, cols, data, step = tables[0]
    assert key == "TrainEpoch/CM" and len(cols) == 10 and sum(map(sum, data)) == 6 and step == 0
    m.training_step((x, y), 0)
    m.on_train_epoch_end()
    assert tables[1][3] == 1 and sum(map(sum, tables[1][2])) == 6          # reset between epochs
