"""Federated Averaging (reference: fedstellar/learning/aggregators/fedavg.py:14-60;
McMahan et al., https://arxiv.org/abs/1602.05629).

``out[layer] = sum_i w_i * m_i[layer] / sum_i w_i``. The reference walks models x layers in Python
on CPU tensors; here the floating-point tensors of every model are packed into one ``[n, P]``
matrix and reduced in ONE pass — by the fused gfx950 kernel (``csrc/fedavg.hip``) when the models
live on the GPU. Integer buffers (BatchNorm ``num_batches_tracked``) are not averaged as floats
(fedavg.py:52-58 does): the maximum is taken.
"""
from __future__ import annotations

import logging
from collections import OrderedDict

import torch

from ...runtime.aggregation import build_plan, run_plan
from .aggregator import Aggregator


class FedAvg(Aggregator):
    def __init__(self, node_name="unknown", config=None):
        super().__init__(node_name, config)

    def aggregate(self, models):
        if len(models) == 0:
            logging.error("[FedAvg] Trying to aggregate models when there are no models")
            return None
        entries = list(models.values())
        ref_sd = entries[0][0]
        keys = list(ref_sd.keys())
        float_keys = [k for k in keys if torch.is_floating_point(ref_sd[k])]
        weights = [float(w) for _, w in entries]
        if sum(weights) <= 0:
            weights = [1.0] * len(entries)
        out = OrderedDict()
        if float_keys:
            dev = ref_sd[float_keys[0]].device
            n = sum(ref_sd[k].numel() for k in float_keys)
            n_pad = -(-n // 4) * 4
            rows = torch.zeros(len(entries), n_pad, dtype=torch.float32, device=dev)
            for i, (sd, _) in enumerate(entries):
                torch.cat([sd[k].reshape(-1).to(torch.float32) for k in float_keys], out=rows[i, :n])
            avg = torch.zeros(1, n_pad, dtype=torch.float32, device=dev)
            run_plan(rows, None, avg, build_plan([(list(range(len(entries))), weights, [0])]))
            off = 0
            for k in float_keys:
                t = ref_sd[k]
                out[k] = avg[0, off:off + t.numel()].reshape(t.shape).to(t.dtype)
                off += t.numel()
        for k in keys:
            if k not in out:   # integer buffers
                out[k] = torch.stack([sd[k] for sd, _ in entries]).max(dim=0).values
        return OrderedDict((k, out[k]) for k in keys)
