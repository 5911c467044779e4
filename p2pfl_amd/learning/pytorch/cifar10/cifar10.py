"""CIFAR-10 data module (reference: fedstellar/learning/pytorch/cifar10/cifar10.py:20-116).

Reads ``cifar10.npz`` (keys ``x_train [50000,32,32,3] uint8, y_train, x_test, y_test``) from
``root_dir`` when present (the python-pickle batches of the original archive are not loaded:
unpickling downloaded files is unsafe), otherwise a synthetic 3x32x32 10-class set. Training
augmentation of the reference (RandomCrop(32, pad 4) + horizontal flip + Normalize,
cifar10.py:62-69) is applied per batch by :meth:`augment`; test == validation set, ``drop_last``
on the training loader (cifar10.py:76-77,115-116).
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..datamodule import FederatedDataModule, synthetic_classification, LOAD_LOCK

MEAN = torch.tensor([0.4914, 0.4822, 0.4465]).view(1, 3, 1, 1)
STD = torch.tensor([0.2470, 0.2435, 0.2616]).view(1, 3, 1, 1)


class CIFAR10DataModule(FederatedDataModule):
    num_classes = 10
    input_shape = (3, 32, 32)
    cifar10_train = None
    cifar10_val = None
    synthetic = False

    def __init__(self, sub_id=0, number_sub=1, batch_size=32, num_workers=4, val_percent=0.1, iid=True,
                 root_dir=None, seed=0, normalization="cifar10", loading="torchvision"):
        super().__init__(sub_id, number_sub, batch_size, num_workers, val_percent, iid, seed)
        if loading != "torchvision":
            raise NotImplementedError(loading)              # as in the reference (cifar10.py:55-58)
        self.normalization, self.loading = normalization, loading
        stats = self.set_normalization(normalization)
        self.mean, self.std = stats["mean"], stats["std"]
        self.root_dir = root_dir or os.path.join(os.getcwd(), "data")
        cls = CIFAR10DataModule
        with LOAD_LOCK:
            if cls.cifar10_train is None:
                cls._load(self.root_dir)
        self._install(*cls.cifar10_train, *cls.cifar10_val, drop_val=True)
        if normalization != "cifar10":                      # the cached arrays carry the CIFAR-10 statistics
            m2, s2 = torch.tensor(self.mean).view(1, 3, 1, 1), torch.tensor(self.std).view(1, 3, 1, 1)
            renorm = lambda t: (((t.reshape(-1, 3, 32, 32) * STD + MEAN) - m2) / s2).reshape(len(t), -1).contiguous()
            self.train_x, self.test_x = renorm(self.train_x), renorm(self.test_x)
        self.val_x, self.val_y = self.test_x, self.test_y   # reference: val loader == test loader

    @staticmethod
    def set_normalization(normalization):
        """Channel statistics by name (reference cifar10.py:32-45)."""
        if normalization == "cifar10":
            return {"mean": (0.4914, 0.4822, 0.4465), "std": (0.2470, 0.2435, 0.2616)}
        if normalization == "imagenet":
            return {"mean": (0.485, 0.456, 0.406), "std": (0.229, 0.224, 0.225)}
        raise NotImplementedError(normalization)

    def get_dataset(self, train, transform=None, download=False):
        """The WHOLE (unpartitioned) split as a dataset of normalised ``[3, 32, 32]`` tensors and labels (reference
        cifar10.py:47-59 hands out torchvision's CIFAR10; nothing is downloaded here - ``cifar10.npz`` under
        ``root_dir`` or the synthetic stand-in, see the module docstring). ``transform``: callable applied per sample."""
        x, y = CIFAR10DataModule.cifar10_train if train else CIFAR10DataModule.cifar10_val
        ds = torch.utils.data.TensorDataset(x.reshape(len(x), 3, 32, 32), y)
        if transform is None:
            return ds

        class _Mapped(torch.utils.data.Dataset):
            def __len__(self_):
                return len(ds)

            def __getitem__(self_, i):
                xi, yi = ds[i]
                return transform(xi), yi
        return _Mapped()

    @classmethod
    def _load(cls, root):
        path = os.path.join(root, "cifar10.npz")
        if os.path.exists(path):
            d = np.load(path)
            prep = lambda a: (torch.from_numpy(a).permute(0, 3, 1, 2).float() / 255.0 - MEAN) / STD
            cls.cifar10_train = (prep(d["x_train"]).reshape(len(d["x_train"]), -1), torch.from_numpy(d["y_train"]).long().flatten())
            cls.cifar10_val = (prep(d["x_test"]).reshape(len(d["x_test"]), -1), torch.from_numpy(d["y_test"]).long().flatten())
            cls.synthetic = False
        else:
            logging.warning("[CIFAR10] %s not found: using the synthetic CIFAR-shaped set", path)
            tx, ty, ex, ey = synthetic_classification(50000, 10000, 3 * 32 * 32, 10, seed=777, binary_images=False)
            cls.cifar10_train, cls.cifar10_val, cls.synthetic = (tx, ty), (ex, ey), True

    @staticmethod
    def augment(x: torch.Tensor, generator=None) -> torch.Tensor:
        """RandomCrop(32, padding=4) + RandomHorizontalFlip on a [B,3,32,32] batch - one gather on the batch's device
        (offsets and flips drawn there too, unless a host ``generator`` is given), no per-sample loop, no host
        round trip."""
        b, dev = x.shape[0], x.device
        draw = dev if generator is None else generator.device
        xp = F.pad(x, (4, 4, 4, 4))
        ox = torch.randint(0, 9, (b,), generator=generator, device=draw).to(dev)
        oy = torch.randint(0, 9, (b,), generator=generator, device=draw).to(dev)
        flip = (torch.rand(b, generator=generator, device=draw) < 0.5).to(dev)
        ar = torch.arange(32, device=dev)
        rows = oy.view(b, 1) + ar.view(1, 32)                                            # [b, 32]
        cols = ox.view(b, 1) + torch.where(flip.view(b, 1), 31 - ar.view(1, 32), ar.view(1, 32))
        bi = torch.arange(b, device=dev).view(b, 1, 1)
        return xp[bi, :, rows.view(b, 32, 1), cols.view(b, 1, 32)].permute(0, 3, 1, 2).contiguous()

    def train_dataloader(self):
        return self._loader(self.train_x, self.train_y, True, drop_last=True)
