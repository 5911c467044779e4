"""SYSCALL (malware system-call frequency) data module
(reference: fedstellar/learning/pytorch/syscall/syscall.py:21-218).

17 scaled 1-gram system-call frequency features, 9 classes, 85/15 train/test split, validation =
1 % of the shard (syscall.py:60-112,139-181). The reference downloads a zip of CSVs (no network here);
looked up under ``root_dir``, in this order:

* ``syscall.pt`` - ``(x_train, y_train, x_test, y_test)`` tensors (``weights_only=True``);
* ``raw/*.csv``  - the reference's raw files: tab-separated, column ``system calls frequency_1gram-scaled``
  holding a Python list literal per row and the class name in ``maltype`` (``normalv2`` is folded into
  ``normal``), processed exactly like ``SYSCALL.process`` (syscall.py:74-112): 85/15
  ``train_test_split(random_state=42)``; classes are numbered in sorted name order (the reference numbers
  them in ``set()`` iteration order, which changes from run to run);
* ``syscall.csv`` - plain numeric CSV, 17 feature columns + label column;
* otherwise a synthetic set of the same shape.
"""
from __future__ import annotations

import ast
import glob
import logging
import os

import torch

from ..datamodule import ArchiveDataset, FederatedDataModule, sort_dataset, synthetic_classification, LOAD_LOCK  # noqa: F401


FEATURE = "system calls frequency_1gram-scaled"


def process_raw_csvs(files):
    """Reference ``SYSCALL.process`` on its raw CSV files -> (x_train, y_train, x_test, y_test, classes)."""
    import csv as _csv
    feats, names = [], []
    for path in files:
        with open(path, newline="") as f:
            for row in _csv.DictReader(f, delimiter="\t"):
                feats.append([float(v) for v in ast.literal_eval(row[FEATURE])])
                names.append("normal" if row["maltype"] == "normalv2" else row["maltype"])
    classes = sorted(set(names))
    index = {c: i for i, c in enumerate(classes)}
    x = torch.tensor(feats, dtype=torch.float32)
    y = torch.tensor([index[n] for n in names], dtype=torch.long)
    from sklearn.model_selection import train_test_split
    itr, ite = train_test_split(list(range(len(x))), test_size=0.15, random_state=42)
    return x[itr], y[itr], x[ite], y[ite], classes


class SYSCALL(ArchiveDataset):
    """The system-call frequency set as a dataset object (reference syscall/syscall.py:21-112): raw tab-separated
    CSVs -> ``processed/syscall_{train,test}.pt`` = ``[x, y, classes_to_targets, classes]`` (85 / 15 split,
    ``random_state=42``; classes numbered in sorted name order, see the module docstring)."""
    FOLDER = "syscall"
    download_link = "https://files.ifi.uzh.ch/CSG/research/fl/data/syscall.zip"

    def _files(self):
        pr = os.path.join(self.base, "processed")
        return [os.path.join(pr, "syscall_train.pt")], [os.path.join(pr, "syscall_test.pt")]

    def process(self):
        files = sorted(glob.glob(os.path.join(self.base, "raw", "**", "*.csv"), recursive=True))
        if not files:
            raise RuntimeError(f"no raw CSV files under {os.path.join(self.base, 'raw')}")
        tx, ty, ex, ey, classes = process_raw_csvs(files)
        index = {c: i for i, c in enumerate(classes)}
        os.makedirs(os.path.join(self.base, "processed"), exist_ok=True)
        (train_file,), (test_file,) = self._files()
        torch.save([tx, ty, index, classes], train_file)
        torch.save([ex, ey, index, classes], test_file)

    def _read(self, train):
        blob = torch.load(self._files()[0 if train else 1][0], weights_only=True)
        self.classes_to_targets, self.classes = (blob[2], blob[3]) if len(blob) > 3 else (None, None)
        return blob[0], blob[1]


class SYSCALLDataModule(FederatedDataModule):
    classes = None
    num_classes = 9
    input_shape = (17,)
    train_set = None
    test_set = None
    synthetic = False

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.01, iid=True,
                 root_dir=None, seed=0):
        super().__init__(sub_id, number_sub, batch_size, num_workers, val_percent, iid, seed)
        cls = SYSCALLDataModule
        with LOAD_LOCK:
            if cls.train_set is None:
                cls._load(root_dir or os.path.join(os.getcwd(), "data", "syscall"))
        self._install(*cls.train_set, *cls.test_set)

    @classmethod
    def _load(cls, root):
        pt, csv = os.path.join(root, "syscall.pt"), os.path.join(root, "syscall.csv")
        processed = os.path.join(root, "processed", "syscall_train.pt")      # written by the SYSCALL dataset object
        if os.path.exists(processed) and os.path.exists(processed.replace("_train", "_test")):
            tr, te = (torch.load(f, weights_only=True) for f in (processed, processed.replace("_train", "_test")))
            tx, ty, ex, ey = tr[0], tr[1], te[0], te[1]
            cls.classes = tr[3] if len(tr) > 3 else None
            cls.synthetic = False
        elif os.path.exists(pt):
            tx, ty, ex, ey = torch.load(pt, weights_only=True)
            cls.synthetic = False
        elif glob.glob(os.path.join(root, "raw", "*.csv")):
            tx, ty, ex, ey, cls.classes = process_raw_csvs(sorted(glob.glob(os.path.join(root, "raw", "*.csv"))))
            cls.synthetic = False
        elif os.path.exists(csv):
            import numpy as np
            a = np.loadtxt(csv, delimiter=",", skiprows=1)
            x, y = torch.from_numpy(a[:, :17]).float(), torch.from_numpy(a[:, 17]).long()
            perm = torch.randperm(len(x), generator=torch.Generator().manual_seed(0))
            cut = int(len(x) * 0.85)
            tx, ty, ex, ey = x[perm[:cut]], y[perm[:cut]], x[perm[cut:]], y[perm[cut:]]
            cls.synthetic = False
        else:
            logging.warning("[SYSCALL] no data under %s: using the synthetic set", root)
            tx, ty, ex, ey = synthetic_classification(17000, 3000, 17, 9, seed=99, noise=0.15, binary_images=False)
            cls.synthetic = True
        cls.train_set, cls.test_set = (tx.float(), ty.long()), (ex.float(), ey.long())
