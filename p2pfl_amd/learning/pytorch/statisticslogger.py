"""Local metric loggers (reference: fedstellar/learning/pytorch/statisticslogger.py:26-243 and
Lightning's ``CSVLogger`` used at node.py:122-125).

* :class:`FedstellarLogger` writes TensorBoard event files (``events.out.tfevents.*``) with an own
  minimal TFRecord/protobuf encoder — tensorboardX is not available on the target image. It keeps
  the reference's federated stepping: a scalar logged at trainer step ``s`` lands at
  ``global_step + s``; ``finalize_round`` (``LightningLearner.finalize_round``) folds the local step
  into the global one so rounds line up on one axis (statisticslogger.py:60-61,132-136).
* :class:`CSVLogger` appends ``step,key,value`` rows to ``<save_dir>/<name>/<version>/metrics.csv``.
"""
from __future__ import annotations

import csv
import os
import socket
import struct
import threading
import time
from typing import Dict, Optional

# ---------------------------------------------------------------------------- crc32c (Castagnoli)
_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---------------------------------------------------------------------------- protobuf (hand encoded)
def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field_bytes(num: int, payload: bytes) -> bytes:
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def _event(wall_time: float, step: int, summary: Optional[bytes] = None, file_version: Optional[str] = None) -> bytes:
    ev = _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(max(step, 0))
    if file_version is not None:
        ev += _field_bytes(3, file_version.encode())
    if summary is not None:
        ev += _field_bytes(5, summary)
    return ev


def _scalar_summary(tag: str, value: float) -> bytes:
    val = _field_bytes(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
    return _field_bytes(1, val)


# ---------------------------------------------------------------------------- reader (statistics page)
def _read_varint(buf: bytes, pos: int):
    n = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, pos


def _pb_fields(buf: bytes):
    """(field number, wire type, value) of one protobuf message: varints as int, 64/32-bit as raw bytes, length-
    delimited as bytes. Enough for Event / Summary / Summary.Value."""
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _read_varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        else:
            return
        yield num, wt, v


def read_event_file(path: str, check_crc: bool = True) -> Dict[str, list]:
    """Scalars of one ``events.out.tfevents.*`` file as ``{tag: [[step, value], ...]}`` - written by this module, by
    TensorBoard's SummaryWriter or by Lightning's TensorBoardLogger (the reference's logger,
    fedstellar/learning/pytorch/statisticslogger.py:13-60): the record framing and the Event / Summary fields are
    the same. A truncated tail (a run still writing) ends the read quietly; a CRC mismatch raises ValueError."""
    out: Dict[str, list] = {}
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    while pos + 12 <= len(data):
        header = data[pos:pos + 8]
        (ln,) = struct.unpack("<Q", header)
        if pos + 12 + ln + 4 > len(data):
            break
        if check_crc and struct.unpack("<I", data[pos + 8:pos + 12])[0] != _masked_crc(header):
            raise ValueError(f"{path}: corrupt record header at byte {pos}")
        rec = data[pos + 12:pos + 12 + ln]
        if check_crc and struct.unpack("<I", data[pos + 12 + ln:pos + 16 + ln])[0] != _masked_crc(rec):
            raise ValueError(f"{path}: corrupt record at byte {pos}")
        pos += 16 + ln
        step, summary = 0, None
        for num, wt, v in _pb_fields(rec):
            if num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for num, wt, val in _pb_fields(summary):
            if num != 1 or wt != 2:
                continue
            tag, x = None, None
            for n2, w2, v2 in _pb_fields(val):
                if n2 == 1 and w2 == 2:
                    tag = v2.decode("utf-8", "replace")
                elif n2 == 2 and w2 == 5:
                    x = struct.unpack("<f", v2)[0]
            if tag is not None and x is not None:
                out.setdefault(tag, []).append([int(step), float(x)])
    return out


class _EventWriter:
    def __init__(self, logdir: str):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.{os.getpid()}")
        self._f = open(self.path, "ab")
        self._lock = threading.Lock()
        self._write(_event(time.time(), 0, file_version="brain.Event:2"))

    def _write(self, data: bytes):
        header = struct.pack("<Q", len(data))
        with self._lock:
            self._f.write(header + struct.pack("<I", _masked_crc(header)) + data + struct.pack("<I", _masked_crc(data)))

    def add_scalar(self, tag: str, value: float, step: int):
        self._write(_event(time.time(), int(step), summary=_scalar_summary(tag, value)))

    def flush(self):
        with self._lock:
            self._f.flush()

    def close(self):
        with self._lock:
            if not self._f.closed:
                self._f.flush()
                self._f.close()


class _FederatedStepping:
    """Shared FL-aware step accounting."""

    def __init__(self):
        self.global_step = 0
        self.local_step = 0

    def _step(self, step):
        if step is not None:
            self.local_step = max(self.local_step, int(step))
        return self.global_step + (int(step) if step is not None else self.local_step)

    def finalize_round(self):
        self.global_step += self.local_step
        self.local_step = 0


class FedstellarLogger(_FederatedStepping):
    """TensorBoard logger: ``<save_dir>/<name>/<version>/events.out.tfevents.*``."""

    def __init__(self, save_dir, name="default", version=None, log_graph=False, sub_dir=None, **kwargs):
        super().__init__()
        self.save_dir, self.name_, self.version_ = os.fspath(save_dir), name or "", version
        self.sub_dir = None if sub_dir is None else os.fspath(sub_dir)
        self._writer = None
        self.hparams = {}

    # directory layout of the reference logger (statisticslogger.py:63-88): <save_dir>/<name>/<version>[/<sub_dir>];
    # an integer (or missing) version becomes "version_<n>", a missing one the next free n
    @property
    def root_dir(self) -> str:
        return os.path.join(self.save_dir, self.name)

    @property
    def log_dir(self) -> str:
        v = self.version
        d = os.path.join(self.root_dir, v if isinstance(v, str) else f"version_{v}")
        if self.sub_dir:
            d = os.path.join(d, self.sub_dir)
        return os.path.expanduser(os.path.expandvars(d))

    @property
    def experiment(self) -> _EventWriter:
        if self._writer is None:
            self._writer = _EventWriter(self.log_dir)
        return self._writer

    @property
    def name(self):
        return self.name_

    @property
    def version(self):
        if self.version_ is None:
            self.version_ = self._get_next_version()
        return self.version_

    def _get_next_version(self) -> int:
        """One more than the highest ``version_<n>`` directory under ``root_dir`` (0 when there is none)."""
        try:
            names = os.listdir(self.root_dir)
        except OSError:
            return 0
        seen = [int(n.split("_", 1)[1]) for n in names
                if n.startswith("version_") and n.split("_", 1)[1].isdigit() and os.path.isdir(os.path.join(self.root_dir, n))]
        return max(seen) + 1 if seen else 0

    @staticmethod
    def _sanitize_params(params) -> dict:
        """Hyper-parameters as plain values: arrays / tensors with more than one dimension are logged as text."""
        import numpy as np
        out = {}
        for k, v in dict(params or {}).items():
            if isinstance(v, np.ndarray) or (hasattr(v, "ndim") and hasattr(v, "shape")):
                if getattr(v, "ndim", 0) > 1:
                    v = str(v)
                elif getattr(v, "ndim", 0) == 0:
                    v = v.item()
                else:
                    v = [x.item() if hasattr(x, "item") else x for x in v]
            elif callable(v):
                v = getattr(v, "__name__", repr(v))
            out[str(k)] = v
        return out

    def __getstate__(self):
        """Picklable (a learner travels to worker processes with its logger): the open event file stays behind."""
        state = self.__dict__.copy()
        state["_writer"] = None
        return state

    def log_metrics(self, metrics: Dict[str, float], step: Optional[int] = None):
        s = self._step(step)
        for k, v in metrics.items():
            try:
                self.experiment.add_scalar(k, float(v), s)
            except (TypeError, ValueError):
                continue

    def log_metrics_direct(self, metrics: Dict[str, float], step: int):
        """Log at an absolute step (resource metrics are indexed by wall-clock seconds)."""
        for k, v in metrics.items():
            self.experiment.add_scalar(k, float(v), int(step))

    def log_hyperparams(self, params, metrics=None):
        self.hparams.update(self._sanitize_params(params))
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "hparams.yaml"), "w") as f:
            for k, v in self.hparams.items():
                f.write(f"{k}: {v}\n")

    def log_graph(self, model, input_array=None):
        """Graph dumps need the TensorBoard graph proto; the architecture is logged as text instead."""
        os.makedirs(self.log_dir, exist_ok=True)
        with open(os.path.join(self.log_dir, "model.txt"), "w") as f:
            f.write(repr(model))

    def save(self):
        if self._writer is not None:
            self._writer.flush()

    def finalize(self, status="success"):
        if self._writer is not None:
            self._writer.close()

    close = finalize


class CSVLogger(_FederatedStepping):
    """``<save_dir>/<name>/<version>/metrics.csv`` with rows ``wall_time,step,key,value``."""

    def __init__(self, save_dir, name="metrics", version=None, **kwargs):
        super().__init__()
        self.log_dir = os.path.join(save_dir, name or "", str(version) if version is not None else "version_0")
        os.makedirs(self.log_dir, exist_ok=True)
        self.path = os.path.join(self.log_dir, "metrics.csv")
        self._lock = threading.Lock()
        new = not os.path.exists(self.path)
        self._f = open(self.path, "a", newline="")
        self._w = csv.writer(self._f)
        if new:
            self._w.writerow(["wall_time", "step", "key", "value"])

    def log_metrics(self, metrics, step=None):
        s = self._step(step)
        with self._lock:
            for k, v in metrics.items():
                try:
                    self._w.writerow([f"{time.time():.3f}", s, k, float(v)])
                except (TypeError, ValueError):
                    continue
            self._f.flush()

    def log_metrics_direct(self, metrics, step):
        with self._lock:
            for k, v in metrics.items():
                self._w.writerow([f"{time.time():.3f}", int(step), k, float(v)])
            self._f.flush()

    def log_hyperparams(self, params, metrics=None):
        with open(os.path.join(self.log_dir, "hparams.yaml"), "w") as f:
            for k, v in (dict(params) if params else {}).items():
                f.write(f"{k}: {v}\n")

    def log_graph(self, model, input_array=None):
        pass

    def save(self):
        with self._lock:
            self._f.flush()

    def finalize(self, status="success"):
        with self._lock:
            if not self._f.closed:
                self._f.close()

    close = finalize
