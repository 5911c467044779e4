"""WADI (water-distribution anomaly detection) data module
(reference: fedstellar/learning/pytorch/wadi/wadi.py:20-163): 123 features, binary label, read from
``X_train.npy / y_train.npy / X_test.npy / y_test.npy`` under ``root_dir`` (the reference's download
URL is a placeholder, wadi.py:25), synthetic stand-in otherwise.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch

from ..datamodule import ArchiveDataset, FederatedDataModule, synthetic_classification, LOAD_LOCK


class WADI(ArchiveDataset):
    """WADI as a dataset object (reference wadi/wadi.py:20-60): ``X_{train,test}.npy`` / ``y_{train,test}.npy`` under
    ``<root_dir>/WADI``. The reference's download location is a placeholder ('XXXX', wadi.py:25): there is nothing to
    fetch, ``download=True`` says where the files have to be placed."""
    FOLDER = "WADI"
    download_link = "XXXX"
    files = ["X_train.npy", "y_train.npy", "X_test.npy", "y_test.npy"]

    def _files(self):
        f = [os.path.join(self.base, n) for n in self.files]
        return f[:2], f[2:]

    def _read(self, train):
        xf, yf = self._files()[0 if train else 1]
        return torch.from_numpy(np.load(xf)).to(torch.float32), torch.from_numpy(np.load(yf)).to(torch.float32)


class WADIDataModule(FederatedDataModule):
    num_classes = 2
    input_shape = (123,)
    train_set = None
    test_set = None
    synthetic = False

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.1, iid=True,
                 root_dir=None, seed=0):
        super().__init__(sub_id, number_sub, batch_size, num_workers, val_percent, iid, seed)
        cls = WADIDataModule
        with LOAD_LOCK:
            if cls.train_set is None:
                cls._load(root_dir or os.path.join(os.getcwd(), "data", "WADI"))
        self._install(*cls.train_set, *cls.test_set)

    @classmethod
    def _load(cls, root):
        names = ["X_train.npy", "y_train.npy", "X_test.npy", "y_test.npy"]
        if all(os.path.exists(os.path.join(root, n)) for n in names):
            tx, ty, ex, ey = (torch.from_numpy(np.load(os.path.join(root, n))) for n in names)
            cls.synthetic = False
        else:
            logging.warning("[WADI] no data under %s: using the synthetic set", root)
            tx, ty, ex, ey = synthetic_classification(20000, 4000, 123, 2, seed=55, noise=0.2, binary_images=False)
            cls.synthetic = True
        cls.train_set, cls.test_set = (tx.float(), ty.long().flatten()), (ex.float(), ey.long().flatten())
