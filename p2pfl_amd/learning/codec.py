"""Serialized model payload == checkpoint format.

Byte-compatible with the reference: ``pickle.dumps((list[np.ndarray] in state_dict order,
contributors: list[str] | None, weight: int | None))`` (lightninglearner.py:73-89). Decoding uses a
*restricted* unpickler: the payload arrives from the network, and a raw ``pickle.loads`` on it
(lightninglearner.py:81) is remote code execution. Only the globals numpy needs to rebuild
ndarrays are resolvable; everything else raises :class:`DecodingParamsError`.
"""
from __future__ import annotations

import io
import json
import os
import pickle
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .exceptions import DecodingParamsError

_ALLOWED = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "str"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"global {module}.{name} is not allowed in a model payload")


def encode_parameters(arrays: Sequence[np.ndarray], contributors: Optional[List[str]] = None,
                      weight: Optional[int] = None) -> bytes:
    # (np.ascontiguousarray would turn 0-d entries - BatchNorm's num_batches_tracked - into shape (1,))
    arrays = [np.asarray(a, order="C") for a in arrays]
    return pickle.dumps((arrays, contributors, weight))


def decode_parameters(data: bytes) -> Tuple[List[np.ndarray], Optional[List[str]], Optional[int]]:
    try:
        obj = _RestrictedUnpickler(io.BytesIO(data)).load()
        arrays, contributors, weight = obj
        arrays = [np.asarray(a) for a in arrays]
        if contributors is not None:
            contributors = [str(c) for c in contributors]
        if weight is not None:
            weight = int(weight)
        return arrays, contributors, weight
    except Exception as e:
        raise DecodingParamsError(str(e)) from e


# ------------------------------------------------------------------ checkpoint files
def save_checkpoint(path: str, arrays: Sequence[np.ndarray], contributors=None, weight=None, meta: dict = None):
    """Write the payload bytes to ``path`` (+ ``path.json`` sidecar with round / role / policy)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(encode_parameters(arrays, contributors, weight))
    os.replace(tmp, path)
    if meta is not None:
        with open(path + ".json", "w") as f:
            json.dump(meta, f, indent=2)


def load_checkpoint(path: str):
    with open(path, "rb") as f:
        arrays, contributors, weight = decode_parameters(f.read())
    meta = None
    if os.path.exists(path + ".json"):
        with open(path + ".json") as f:
            meta = json.load(f)
    return arrays, contributors, weight, meta
