"""Classification metrics from a confusion matrix.

The reference logs torchmetrics ``Multiclass{Accuracy,Precision,Recall,F1Score}`` with their
defaults (``average="macro"``) plus a ``MulticlassConfusionMatrix`` per phase
(mnist/models/mlp.py:118-132). torchmetrics is not available on the target image, and the engine
produces the confusion matrix on the GPU anyway (``csrc/loss.hip``), so the four scalars are derived
from it with the same definitions: per-class statistics, macro average over the classes that
occur (as target or prediction) — torchmetrics gives absent classes zero weight.
"""
from __future__ import annotations

from typing import Dict

import torch


def metrics_from_confusion(cm: torch.Tensor) -> Dict[str, float]:
    """cm [C, C], rows = target, cols = prediction."""
    cm = cm.to(torch.float64)
    tp = cm.diag()
    fp = cm.sum(0) - tp
    fn = cm.sum(1) - tp
    present = (tp + fp + fn) > 0
    n = max(int(present.sum()), 1)

    def safe(num, den):
        return torch.where(den > 0, num / den.clamp(min=1e-30), torch.zeros_like(num))

    precision = safe(tp, tp + fp)
    recall = safe(tp, tp + fn)
    f1 = safe(2 * tp, 2 * tp + fp + fn)
    w = present.to(torch.float64)
    return {
        "Accuracy": float((recall * w).sum() / n),       # macro accuracy == mean per-class recall
        "Precision": float((precision * w).sum() / n),
        "Recall": float((recall * w).sum() / n),
        "F1Score": float((f1 * w).sum() / n),
        "MicroAccuracy": float(tp.sum() / cm.sum().clamp(min=1)),
    }
