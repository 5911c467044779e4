"""Synthetic MNIST / FEMNIST-shaped shards (there is no network for the real datasets).

Images are class prototypes plus noise in [0, 1] (28x28, one channel, like ``ToTensor`` MNIST,
reference mnist/mnist.py:72-75), so models genuinely learn and accuracy curves are meaningful.
Shard sizes follow the reference partitioning: node i of N gets a contiguous block of
``floor(60000 / N)`` training rows split 90/10 into train/validation and ``floor(10000 / N)`` test
rows (mnist/mnist.py:100-118).
"""
from __future__ import annotations

from typing import Tuple

import torch


def reference_shard_sizes(n_nodes: int, n_train_total: int = 60000, n_test_total: int = 10000,
                          val_percent: float = 0.1) -> Tuple[int, int, int]:
    rows = n_train_total // n_nodes
    n_tr = round(rows * (1 - val_percent))
    n_va = round(rows * val_percent)
    return n_tr, n_va, n_test_total // n_nodes


def synthetic_shards(n_clients: int, n_train: int, n_val: int, n_test: int, num_classes: int = 10,
                     seed: int = 0, device="cpu", noise: float = 0.35, first_client: int = 0):
    """Returns (train_x, train_y, val_x, val_y, test_x, test_y); x: [K, n, 784] fp32, y: [K, n] int32.

    The class prototypes depend only on ``seed`` (every rank sees the same task); the samples of
    client k depend on ``seed`` and its global index ``first_client + k``."""
    dev = torch.device(device)
    g = torch.Generator(device="cpu").manual_seed(seed)
    protos = (torch.rand(num_classes, 784, generator=g) < 0.25).float() * torch.rand(num_classes, 784, generator=g)
    protos = protos.to(dev)

    def make(n, salt):
        if n == 0:
            return None, None
        xs, ys = [], []
        for k in range(n_clients):
            gk = torch.Generator(device="cpu").manual_seed(seed * 1000003 + (first_client + k) * 7919 + salt)
            y = torch.randint(0, num_classes, (n,), generator=gk)
            nz = torch.randn(n, 784, generator=gk) * noise
            xs.append(nz)
            ys.append(y)
        y = torch.stack(ys).to(dev)
        x = (protos[y] + torch.stack(xs).to(dev)).clamp_(0.0, 1.0)
        return x, y.to(torch.int32)

    tx, ty = make(n_train, 1)
    vx, vy = make(n_val, 2)
    ex, ey = make(n_test, 3)
    return tx, ty, vx, vy, ex, ey
