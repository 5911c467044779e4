"""MNIST data module (reference: fedstellar/learning/pytorch/mnist/mnist.py:36-158).

Loads the IDX files (``train-images-idx3-ubyte[.gz]`` ...) from ``root_dir`` / ``<cwd>/data/MNIST/raw``
when they exist; there is no network on the target machines, so otherwise a deterministic synthetic
MNIST-shaped set (60 000 / 10 000 samples, 10 classes, pixel range [0, 1]) is generated once per
process and shared by every node (the reference keeps the same class-level singleton,
mnist/mnist.py:49-50,72-95).
"""
from __future__ import annotations

import gzip
import logging
import os
import struct

import numpy as np
import torch

from ..datamodule import FederatedDataModule, synthetic_classification, LOAD_LOCK


def _read_idx(path: str) -> np.ndarray:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        magic, = struct.unpack(">I", f.read(4))
        ndim = magic & 0xFF
        dims = struct.unpack(">" + "I" * ndim, f.read(4 * ndim))
        return np.frombuffer(f.read(), dtype=np.uint8).reshape(dims)


def _find(root: str, stem: str):
    for d in (root, os.path.join(root, "MNIST", "raw"), os.path.join(root, "raw")):
        for ext in ("", ".gz"):
            p = os.path.join(d, stem + ext)
            if os.path.exists(p):
                return p
    return None


class MNISTDataModule(FederatedDataModule):
    num_classes = 10
    input_shape = (1, 28, 28)
    mnist_train = None   # (x [60000, 784] fp32 in [0,1], y [60000] int64) shared by all nodes
    mnist_val = None
    synthetic = False

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.1, iid=True,
                 root_dir=None, seed=0):
        super().__init__(sub_id, number_sub, batch_size, num_workers, val_percent, iid, seed)
        cls = MNISTDataModule
        with LOAD_LOCK:
            if cls.mnist_train is None or cls.mnist_val is None:
                cls._load(root_dir or os.path.join(os.getcwd(), "data"), seed)
        self._install(*cls.mnist_train, *cls.mnist_val)

    @classmethod
    def _load(cls, root, seed):
        paths = [_find(root, s) for s in ("train-images-idx3-ubyte", "train-labels-idx1-ubyte",
                                          "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")]
        if all(paths):
            tx, ty, ex, ey = (_read_idx(p) for p in paths)
            to_x = lambda a: torch.from_numpy(a.reshape(len(a), -1).astype(np.float32) / 255.0)
            to_y = lambda a: torch.from_numpy(a.astype(np.int64))
            cls.mnist_train, cls.mnist_val, cls.synthetic = (to_x(tx), to_y(ty)), (to_x(ex), to_y(ey)), False
        else:
            logging.warning("[MNIST] IDX files not found under %s: using the synthetic MNIST-shaped set", root)
            # (P2PFL_SYNTHETIC_MNIST_ROWS: a smaller stand-in, e.g. for launcher tests that train on the CPU)
            n_tr = int(os.environ.get("P2PFL_SYNTHETIC_MNIST_ROWS", "60000"))
            tx, ty, ex, ey = synthetic_classification(n_tr, max(n_tr // 6, 100), 784, 10, seed=1234)
            cls.mnist_train, cls.mnist_val, cls.synthetic = (tx, ty), (ex, ey), True
