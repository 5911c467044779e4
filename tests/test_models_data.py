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
