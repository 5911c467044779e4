"""``[clients, P]`` parameter slabs.

All model state of the co-resident federation nodes of one GPU lives in a few dense fp32
matrices (parameters, gradients, optimizer moments): row = client, columns = the model's
tensors in ``state_dict`` order, each tensor starting on a 256-byte boundary, rows padded to a
multiple of 256 bytes. "Sending a model" to a co-resident neighbour is passing a row index;
FedAvg is a weighted row reduce; the reference payload (list of numpy arrays in state_dict
order, ``lightninglearner.py:73-89``) is a strided copy out of a row.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np
import torch

from ..models.specs import ModelSpec

ALIGN = 64  # floats (256 B)


@dataclass(frozen=True)
class ParamSlot:
    name: str
    shape: tuple
    offset: int
    numel: int


class ParamLayout:
    """Column layout of one slab row for a :class:`ModelSpec`."""

    def __init__(self, spec: ModelSpec):
        self.spec = spec
        slots = []
        off = 0
        for p in spec.params:
            slots.append(ParamSlot(p.name, tuple(p.shape), off, p.numel))
            off += -(-p.numel // ALIGN) * ALIGN
        self.slots: List[ParamSlot] = slots
        self.by_name: Dict[str, ParamSlot] = {s.name: s for s in slots}
        self.row_stride = off          # floats per row (multiple of ALIGN)
        self.num_params = spec.num_params

    # ------------------------------------------------------------------ views
    def view(self, slab: torch.Tensor, name: str) -> torch.Tensor:
        """``[K, *shape]`` strided view of parameter ``name`` for all clients."""
        s = self.by_name[name]
        return slab[:, s.offset:s.offset + s.numel].unflatten(1, s.shape)

    def view2d(self, slab: torch.Tensor, name: str) -> torch.Tensor:
        """``[K, shape[0], prod(shape[1:])]`` view (weights as matrices)."""
        s = self.by_name[name]
        rest = s.numel // s.shape[0]
        return slab[:, s.offset:s.offset + s.numel].unflatten(1, (s.shape[0], rest))

    def alloc(self, n_clients: int, device, zero: bool = True) -> torch.Tensor:
        f = torch.zeros if zero else torch.empty
        return f((n_clients, self.row_stride), dtype=torch.float32, device=device)

    # ------------------------------------------------------------------ pack / unpack
    def row_to_arrays(self, row: torch.Tensor) -> List[np.ndarray]:
        """One slab row -> list of numpy arrays in state_dict order (payload format)."""
        host = row.detach().to("cpu")
        return [host[s.offset:s.offset + s.numel].reshape(s.shape).numpy().copy() for s in self.slots]

    def arrays_to_row(self, arrays: Sequence, row: torch.Tensor) -> None:
        if len(arrays) != len(self.slots):
            raise ValueError(f"expected {len(self.slots)} arrays, got {len(arrays)}")
        staged = torch.zeros(self.row_stride, dtype=torch.float32)
        for s, a in zip(self.slots, arrays):
            t = torch.as_tensor(np.asarray(a), dtype=torch.float32)
            if tuple(t.shape) != s.shape:
                raise ValueError(f"{s.name}: expected shape {s.shape}, got {tuple(t.shape)}")
            staged[s.offset:s.offset + s.numel] = t.reshape(-1)
        row.copy_(staged.to(row.device))

    def row_to_state_dict(self, row: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict((s.name, row[s.offset:s.offset + s.numel].reshape(s.shape).clone()) for s in self.slots)

    def state_dict_to_row(self, sd, row: torch.Tensor) -> None:
        """Pack a state_dict into a slab row. Tensors that already live on the row's device are copied there
        directly (device-side pack: no host hop, no numpy); anything else goes through the host path."""
        if all(isinstance(sd[s.name], torch.Tensor) and sd[s.name].device == row.device for s in self.slots):
            for s in self.slots:
                t = sd[s.name].detach()
                if tuple(t.shape) != s.shape:
                    raise ValueError(f"{s.name}: expected shape {s.shape}, got {tuple(t.shape)}")
                row[s.offset:s.offset + s.numel].copy_(t.reshape(-1))
            return
        self.arrays_to_row([sd[s.name].detach().cpu().numpy() for s in self.slots], row)

    def check_state_dict(self, sd) -> bool:
        """Key set and per-tensor shape equality (reference ``check_parameters``,
        lightninglearner.py:91-99)."""
        if set(sd.keys()) != set(self.by_name):
            return False
        return all(tuple(sd[s.name].shape) == s.shape for s in self.slots)

    def layout_hash(self) -> str:
        import hashlib
        h = hashlib.sha1()
        for s in self.slots:
            h.update(f"{s.name}:{s.shape};".encode())
        return h.hexdigest()
