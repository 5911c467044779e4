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
