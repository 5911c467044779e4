"""Tensor-backed federated data modules.

The reference data modules are Lightning ``LightningDataModule``s over torchvision datasets with
4 DataLoader workers per node (e.g. mnist/mnist.py:36-158). Here a node's shard is a pair of
dense tensors (it is tiny: 540 x 784 floats for a 100-node MNIST split) that the shard engine
keeps resident in HBM; ``train/val/test_dataloader()`` are still offered (plain
``torch.utils.data.DataLoader`` over a ``TensorDataset``) for generic torch models.

Partitioning follows the reference: node ``sub_id`` of ``number_sub`` gets the contiguous block
``[sub_id * floor(N/number_sub), (sub_id+1) * floor(N/number_sub))`` of the (optionally label-sorted
= non-IID) training set, split ``1 - val_percent`` / ``val_percent`` at random into train / validation,
and the same contiguous block of the test set (mnist/mnist.py:100-118).
"""
from __future__ import annotations

import logging
import threading
from math import floor
from typing import Tuple

import torch
from torch.utils.data import DataLoader, TensorDataset


# the data modules cache the full dataset on their class: one loader at a time (nodes of one process are built
# from several threads - gateways, tests - and a second concurrent load replaced tensors another thread was slicing)
LOAD_LOCK = threading.RLock()

class FederatedDataModule:
    num_classes = 10
    input_shape = (1, 28, 28)

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.1, iid=True,
                 seed: int = 0):
        if sub_id + 1 > number_sub:
            raise ValueError(f"Not exist the subset {sub_id}")
        self.sub_id, self.number_sub = sub_id, number_sub
        self.batch_size, self.num_workers, self.val_percent = batch_size, num_workers, val_percent
        self.iid, self.seed = iid, seed
        self.train_x = self.train_y = self.val_x = self.val_y = self.test_x = self.test_y = None

    # ------------------------------------------------------------------ sharding helpers
    def _shard(self, x: torch.Tensor, y: torch.Tensor, sort_labels: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        if sort_labels:
            order = torch.sort(y, stable=True)[1]
            x, y = x[order], y[order]
        rows = floor(len(x) / self.number_sub)
        if rows == 0:
            raise ValueError("Too much partitions")
        lo = self.sub_id * rows
        return x[lo:lo + rows], y[lo:lo + rows]

    def _install(self, train_x, train_y, test_x, test_y, drop_val: bool = False):
        tx, ty = self._shard(train_x, train_y, not self.iid)
        ex, ey = self._shard(test_x, test_y, not self.iid)
        n = len(tx)
        n_val = 0 if drop_val else round(n * self.val_percent)
        g = torch.Generator().manual_seed(self.seed * 7919 + self.sub_id)
        perm = torch.randperm(n, generator=g)
        tr, va = perm[:n - n_val], perm[n - n_val:]
        flat = lambda t: t.reshape(len(t), -1).float().contiguous()
        self.train_x, self.train_y = flat(tx[tr]), ty[tr].contiguous()
        self.val_x, self.val_y = (flat(tx[va]), ty[va].contiguous()) if n_val else (None, None)
        self.test_x, self.test_y = flat(ex), ey.contiguous()
        logging.info("Train: %d Val:%d Test:%d", len(self.train_x), n_val, len(self.test_x))

    # ------------------------------------------------------------------ tensor access (engine path)
    def tensors(self):
        """(train_x, train_y, val_x, val_y, test_x, test_y): x [n, features] fp32, y [n]."""
        return self.train_x, self.train_y, self.val_x, self.val_y, self.test_x, self.test_y

    def num_samples(self):
        return len(self.train_x), len(self.test_x)

    # ------------------------------------------------------------------ DataLoader access (generic path)
    def _loader(self, x, y, shuffle, drop_last=False):
        if x is None:
            return None
        ds = TensorDataset(x.reshape(len(x), *self.input_shape), y)
        return DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle, drop_last=drop_last, num_workers=0)

    def train_dataloader(self):
        return self._loader(self.train_x, self.train_y, True)

    def val_dataloader(self):
        return self._loader(self.val_x, self.val_y, False)

    def test_dataloader(self):
        return self._loader(self.test_x, self.test_y, False)


class ArchiveDataset(torch.utils.data.Dataset):
    """Dataset OBJECT of the reference's data files (``FEMNIST`` / ``SYSCALL`` / ``WADI`` classes of
    femnist/femnist.py:24-82, syscall/syscall.py:21-120, wadi/wadi.py:20-60): ``data`` / ``targets`` tensors of the
    train or test split read from ``<root_dir>/<FOLDER>/...``, indexable as ``(sample, int(label))`` with optional
    transforms. Missing files: ``download=True`` fetches ``download_link`` (checksum-verified when ``file_md5`` is set,
    archive members confined to the raw directory) and runs :meth:`process`; otherwise ``RuntimeError`` like the
    reference. The data MODULES of this package read the same files (and fall back to synthetic data without them)."""

    FOLDER = ""
    download_link = ""
    file_md5 = None

    def __init__(self, sub_id=0, number_sub=1, root_dir=None, train=True, transform=None, target_transform=None,
                 download=False):
        self.sub_id, self.number_sub = sub_id, number_sub
        self.root = root_dir if root_dir is not None else "./data"
        self.train, self.transform, self.target_transform, self.download = train, transform, target_transform, download
        if not self._present():
            if not download:
                raise RuntimeError("Dataset not found, set parameter download=True to download")
            self.dataset_download()
            self.process()
        self.data, self.targets = self._read(train)

    # -- to be provided by the dataset ------------------------------------------------------------------------
    def _files(self):
        """(train file(s), test file(s)) that must exist under ``self.base``."""
        raise NotImplementedError

    def _read(self, train):
        raise NotImplementedError

    def process(self):
        """raw -> processed files (nothing to do for archives that ship processed files)."""

    # -- common -----------------------------------------------------------------------------------------------
    @property
    def base(self):
        import os
        return os.path.join(self.root, self.FOLDER)

    def _present(self):
        import os
        return all(os.path.exists(f) for group in self._files() for f in group)

    def dataset_download(self):
        """Fetch ``download_link`` into ``<base>/raw`` and unpack it there (tar / zip; plain files are kept as they
        are). Any URL scheme of ``urllib`` works, ``file://`` included."""
        import hashlib
        import os
        import shutil
        import tarfile
        import urllib.request
        import zipfile
        if not self.download_link or "XXXX" in self.download_link:
            raise RuntimeError(f"{type(self).__name__}: no download location is published for this dataset; "
                               f"place its files under {self.base}")
        raw = os.path.join(self.base, "raw")
        os.makedirs(raw, exist_ok=True)
        os.makedirs(os.path.join(self.base, "processed"), exist_ok=True)
        dst = os.path.join(raw, self.download_link.rstrip("/").split("/")[-1])
        if not os.path.exists(dst):
            with urllib.request.urlopen(self.download_link) as r, open(dst + ".part", "wb") as f:
                shutil.copyfileobj(r, f)
            os.replace(dst + ".part", dst)
        if self.file_md5:
            h = hashlib.md5()
            with open(dst, "rb") as f:
                for chunk in iter(lambda: f.read(1 << 20), b""):
                    h.update(chunk)
            if h.hexdigest() != self.file_md5:
                os.remove(dst)
                raise RuntimeError(f"{dst}: checksum mismatch")
        inside = lambda name: os.path.realpath(os.path.join(raw, name)).startswith(os.path.realpath(raw) + os.sep)
        if tarfile.is_tarfile(dst):
            with tarfile.open(dst) as t:
                members = [m for m in t.getmembers() if (m.isfile() or m.isdir()) and inside(m.name)]
                t.extractall(raw, members=members)
        elif zipfile.is_zipfile(dst):
            with zipfile.ZipFile(dst) as z:
                z.extractall(raw, members=[n for n in z.namelist() if inside(n)])

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        x, target = self.data[index], int(self.targets[index])
        if self.transform is not None:
            x = self.transform(x)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return x, target


def sort_dataset(dataset):
    """Order a dataset object by label in place (the reference's non-IID preparation, syscall/syscall.py:115-119)."""
    order = torch.sort(dataset.targets, stable=True)[1]
    dataset.targets = dataset.targets[order]
    dataset.data = dataset.data[order]
    return dataset


def synthetic_classification(n_train: int, n_test: int, features: int, num_classes: int, seed: int,
                             noise: float = 0.35, binary_images: bool = True):
    """Deterministic learnable stand-in for a dataset that cannot be downloaded (no network)."""
    g = torch.Generator().manual_seed(seed)
    protos = torch.rand(num_classes, features, generator=g)
    if binary_images:
        protos = (torch.rand(num_classes, features, generator=g) < 0.25).float() * protos

    def make(n):
        y = torch.randint(0, num_classes, (n,), generator=g)
        x = protos[y] + noise * torch.randn(n, features, generator=g)
        return (x.clamp_(0, 1) if binary_images else x), y
    return make(n_train) + make(n_test)
