"""FEMNIST data module (reference: fedstellar/learning/pytorch/femnist/femnist.py:24-197).

The reference downloads the LEAF-preprocessed ``femnist_{train,test}.pt`` archives; here they are
read from ``root_dir`` when present (``torch.load(..., weights_only=True)`` -> ``(data, targets)``),
otherwise a synthetic 62-class 28x28 set stands in. The reference's transform chain
CenterCrop(96) -> Grayscale -> Resize(28) -> ColorJitter(contrast=3) -> ToTensor ->
Normalize(0.1307, 0.3081) (femnist.py:125-133) is reproduced with tensor ops (zero-pad to 96,
bilinear resize to 28, optional random contrast, normalise).
"""
from __future__ import annotations

import logging
import os

import torch
import torch.nn.functional as F

from ..datamodule import ArchiveDataset, FederatedDataModule, synthetic_classification, LOAD_LOCK


def femnist_transform(x: torch.Tensor, contrast_jitter: bool = False, generator=None) -> torch.Tensor:
    """x [N, 28, 28] in [0, 1] -> [N, 784] normalised."""
    x = x.reshape(-1, 1, 28, 28).float()
    pad = (96 - 28) // 2
    x = F.pad(x, (pad, pad, pad, pad))
    x = F.interpolate(x, size=(28, 28), mode="bilinear", align_corners=False, antialias=True)
    if contrast_jitter:
        f = torch.empty(len(x), 1, 1, 1).uniform_(0.0, 4.0, generator=generator)
        mean = x.mean(dim=(1, 2, 3), keepdim=True)
        x = ((x - mean) * f + mean).clamp_(0, 1)
    return ((x - 0.1307) / 0.3081).reshape(len(x), -1)


class FEMNIST(ArchiveDataset):
    """LEAF FEMNIST as a dataset object (reference femnist/femnist.py:24-82): ``processed/femnist_{train,test}.pt``
    = ``(data [N,28,28], targets [N])``; the archive ships them processed."""
    FOLDER = "FEMNIST"
    download_link = "https://media.githubusercontent.com/media/GwenLegate/femnist-dataset-PyTorch/main/femnist.tar.gz"
    file_md5 = "60433bc62a9bff266244189ad497e2d7"

    def _files(self):
        pr = os.path.join(self.base, "processed")
        return [os.path.join(pr, "femnist_train.pt")], [os.path.join(pr, "femnist_test.pt")]

    def process(self):
        import shutil
        for name in ("femnist_train.pt", "femnist_test.pt"):
            for dirpath, _, files in os.walk(os.path.join(self.base, "raw")):
                if name in files:
                    shutil.move(os.path.join(dirpath, name), os.path.join(self.base, "processed", name))
                    break

    def _read(self, train):
        data, targets = torch.load(self._files()[0 if train else 1][0], weights_only=True)[:2]
        return data, targets


class FEMNISTDataModule(FederatedDataModule):
    num_classes = 62
    input_shape = (1, 28, 28)
    femnist_train = None
    femnist_val = None
    synthetic = False

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.1, iid=True,
                 root_dir=None, seed=0, contrast_jitter=False):
        super().__init__(sub_id, number_sub, batch_size, num_workers, val_percent, iid, seed)
        cls = FEMNISTDataModule
        with LOAD_LOCK:
            if cls.femnist_train is None or cls.femnist_val is None:
                cls._load(root_dir or os.path.join(os.getcwd(), "data", "FEMNIST"), contrast_jitter)
        self._install(*cls.femnist_train, *cls.femnist_val)

    @classmethod
    def _load(cls, root, contrast_jitter):
        tr, te = os.path.join(root, "femnist_train.pt"), os.path.join(root, "femnist_test.pt")
        if not os.path.exists(tr):             # the reference's layout: <root_dir>/FEMNIST/processed/femnist_*.pt
            for sub in ("processed", os.path.join("FEMNIST", "processed")):
                if os.path.exists(os.path.join(root, sub, "femnist_train.pt")):
                    tr, te = os.path.join(root, sub, "femnist_train.pt"), os.path.join(root, sub, "femnist_test.pt")
                    break
        if os.path.exists(tr) and os.path.exists(te):
            (tx, ty), (ex, ey) = torch.load(tr, weights_only=True)[:2], torch.load(te, weights_only=True)[:2]
            tx, ex = tx.float(), ex.float()
            if tx.max() > 1.5:
                tx, ex = tx / 255.0, ex / 255.0
            cls.synthetic = False
        else:
            logging.warning("[FEMNIST] %s not found: using the synthetic FEMNIST-shaped set", tr)
            tx, ty, ex, ey = synthetic_classification(62000, 6200, 784, 62, seed=4321)
            cls.synthetic = True
        cls.femnist_train = (femnist_transform(tx, contrast_jitter), ty.long())
        cls.femnist_val = (femnist_transform(ex, False), ey.long())
