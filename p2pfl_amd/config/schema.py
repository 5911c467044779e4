"""Typed, validated view of a participant document.

The runtime keeps reading the participant config as the reference does - a mutable dict indexed by strings
(fedstellar/config/config.py:20-109; ``config.participant["GOSSIP_MODELS_FREC"]``, node.py:807), which the heartbeater
and the leadership transfer write into. This module is the checked side of it: ``ParticipantView.from_dict``
coerces every known field to its type, applies the defaults of ``participant.json.example``, rejects values the runtime
could not act on, and keeps unknown keys (``extra``) so documents written by other tools survive ``to_dict``.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field, fields
from typing import Any, Dict, List, Optional, Tuple, Union

FEDERATIONS = ("DFL", "SDFL", "CFL")
ROLES = ("trainer", "aggregator", "proxy", "idle", "server")
ACCELERATORS = ("auto", "cpu", "gpu", "cuda")


def _bool(v) -> bool:
    if isinstance(v, str):
        return v.strip().lower() in ("1", "true", "yes", "on")
    return bool(v)


def _opt_int(v) -> Optional[int]:
    return None if v in (None, "", "auto") else int(v)


@dataclass
class ScenarioArgs:
    name: str = ""
    start_time: str = ""
    federation: str = "DFL"
    rounds: int = 3
    simulation: bool = True
    controller: str = "127.0.0.1:5000"
    random_seed: int = 42
    n_nodes: int = 0
    config_version: str = "development"
    debug: bool = True

    def check(self):
        if self.federation not in FEDERATIONS:
            raise ValueError(f"scenario_args.federation: unknown federation {self.federation!r}")
        if self.rounds < 0 or self.n_nodes < 0:
            raise ValueError("scenario_args: rounds and n_nodes cannot be negative")


@dataclass
class DeviceArgs:
    uid: str = ""
    idx: Optional[int] = None
    username: str = "pi"
    password: str = "pi"
    role: str = "aggregator"
    start: bool = False
    accelerator: str = "auto"
    devices: int = 2
    strategy: str = "ddp"
    gpu: Optional[int] = None            # extension: device index of a one-process-per-node learner
    placement: Optional[int] = None      # extension: rank (GPU) that hosts the node in the batched runtime; None = anywhere

    def check(self):
        if self.role not in ROLES:
            raise ValueError(f"device_args.role: unknown role {self.role!r}")
        if self.accelerator not in ACCELERATORS:
            raise ValueError(f"device_args.accelerator: {self.accelerator!r} is none of {ACCELERATORS}")
        for name in ("gpu", "placement"):
            v = getattr(self, name)
            if v is not None and v < 0:
                raise ValueError(f"device_args.{name} cannot be negative")


@dataclass
class GeoArgs:
    latitude: Union[float, str] = ""
    longitude: Union[float, str] = ""

    def check(self):
        for name, lim in (("latitude", 90.0), ("longitude", 180.0)):
            v = getattr(self, name)
            if v != "" and not -lim <= float(v) <= lim:
                raise ValueError(f"geo_args.{name} {v!r} outside [-{lim}, {lim}]")


@dataclass
class NetworkArgs:
    ip: str = "127.0.0.1"
    ipdemo: str = ""
    port: int = 0
    neighbors: str = ""
    interface: str = "eth0"
    rate: str = "1Mbps"
    loss: str = "0.1%"
    delay: str = "100ms"
    delay_distro: int = 0                # "delay-distro" in the document

    def check(self):
        if not 0 <= self.port < 65536:
            raise ValueError(f"network_args.port {self.port} is not a TCP port")
        for nb in self.neighbor_list():
            host, _, port = nb.rpartition(":")
            if not host or not port.isdigit():
                raise ValueError(f"network_args.neighbors: {nb!r} is not host:port")

    def neighbor_list(self) -> List[str]:
        return self.neighbors.split()


@dataclass
class TrackingArgs:
    enable_remote_tracking: bool = False
    local_tracking: str = "csv"
    log_dir: str = ""
    config_dir: str = ""
    wandb_apikey: str = ""
    wandb_username: str = ""
    wandb_project: str = ""
    wandb_entity: str = ""

    def check(self):
        if self.local_tracking not in ("csv", "web", "tensorboard", "none", ""):
            raise ValueError(f"tracking_args.local_tracking: {self.local_tracking!r} is neither 'csv' nor 'web'")


@dataclass
class Tunables:
    BLOCK_SIZE: int = 2048
    NODE_TIMEOUT: float = 20
    VOTE_TIMEOUT: float = 60
    AGGREGATION_TIMEOUT: float = 60
    HEARTBEAT_PERIOD: float = 4
    HEARTBEATER_REFRESH_NEIGHBORS_BY_PERIOD: int = 4
    WAIT_HEARTBEATS_CONVERGENCE: float = 10
    TRAIN_SET_SIZE: int = 10
    TRAIN_SET_CONNECT_TIMEOUT: float = 5
    AMOUNT_LAST_MESSAGES_SAVED: int = 100
    GOSSIP_MESSAGES_FREC: float = 100
    GOSSIP_MESSAGES_PER_ROUND: int = 100
    GOSSIP_EXIT_ON_X_EQUAL_ROUNDS: int = 20
    GOSSIP_MODELS_FREC: float = 1
    GOSSIP_MODELS_PER_ROUND: int = 2

    def check(self):
        for f in fields(self):
            if getattr(self, f.name) <= 0:
                raise ValueError(f"{f.name} must be a positive number, got {getattr(self, f.name)!r}")
        if self.NODE_TIMEOUT <= self.HEARTBEAT_PERIOD:
            raise ValueError("NODE_TIMEOUT must exceed HEARTBEAT_PERIOD (every node would be evicted between two beats)")


_SECTIONS: Tuple[Tuple[str, type], ...] = (("scenario_args", ScenarioArgs), ("device_args", DeviceArgs),
                                           ("geo_args", GeoArgs), ("network_args", NetworkArgs),
                                           ("tracking_args", TrackingArgs))
_RENAME = {"delay-distro": "delay_distro"}


def _build(cls, doc: Dict[str, Any], where: str):
    kw, extra = {}, {}
    known = {f.name: f for f in fields(cls)}
    for k, v in (doc or {}).items():
        name = _RENAME.get(k, k)
        f = known.get(name)
        if f is None:
            extra[k] = v
            continue
        t = f.type if isinstance(f.type, str) else getattr(f.type, "__name__", str(f.type))
        conv = {"bool": _bool, "int": int, "float": float, "str": str, "Optional[int]": _opt_int}.get(t)
        try:
            kw[name] = v if conv is None or (v is None and t.startswith("Optional")) else conv(v)
        except (TypeError, ValueError):
            raise ValueError(f"{where}.{k}: {v!r} is not {t}") from None
    obj = cls(**kw)
    obj.check()
    return obj, extra


@dataclass
class ParticipantView:
    scenario: ScenarioArgs = field(default_factory=ScenarioArgs)
    device: DeviceArgs = field(default_factory=DeviceArgs)
    geo: GeoArgs = field(default_factory=GeoArgs)
    network: NetworkArgs = field(default_factory=NetworkArgs)
    tracking: TrackingArgs = field(default_factory=TrackingArgs)
    tunables: Tunables = field(default_factory=Tunables)
    dataset: str = "MNIST"
    model: str = "MLP"
    epochs: int = 3
    algorithm: str = "FedAvg"
    extra: Dict[str, Any] = field(default_factory=dict)      # everything this view does not know, kept verbatim

    @classmethod
    def from_dict(cls, doc: Dict[str, Any]) -> "ParticipantView":
        out = cls()
        extra: Dict[str, Any] = {}
        attr = {"scenario_args": "scenario", "device_args": "device", "geo_args": "geo", "network_args": "network",
                "tracking_args": "tracking"}
        for key, typ in _SECTIONS:
            obj, rest = _build(typ, doc.get(key, {}), key)
            setattr(out, attr[key], obj)
            if rest:
                extra[key] = rest
        out.tunables, _ = _build(Tunables, {k: v for k, v in doc.items() if k.isupper() and k in
                                            {f.name for f in fields(Tunables)}}, "participant")
        out.dataset = str(doc.get("data_args", {}).get("dataset", out.dataset))
        out.model = str(doc.get("model_args", {}).get("model", out.model))
        out.epochs = int(doc.get("training_args", {}).get("epochs", out.epochs))
        out.algorithm = str(doc.get("aggregator_args", {}).get("algorithm", out.algorithm))
        if out.epochs < 1:
            raise ValueError("training_args.epochs must be >= 1")
        if out.algorithm != "FedAvg":
            raise ValueError(f"aggregator_args.algorithm: only FedAvg exists (reference node.py:93-99), got {out.algorithm!r}")
        known = set(attr) | {"data_args", "model_args", "training_args", "aggregator_args"} | \
            {f.name for f in fields(Tunables)}
        for k, v in doc.items():
            if k not in known:
                extra[k] = v
            elif k in ("data_args", "model_args", "training_args", "aggregator_args"):
                first = {"data_args": "dataset", "model_args": "model", "training_args": "epochs",
                         "aggregator_args": "algorithm"}[k]
                rest = {a: b for a, b in v.items() if a != first}
                if rest:
                    extra[k] = rest
        out.extra = extra
        return out

    def to_dict(self) -> Dict[str, Any]:
        def sec(obj):
            d = dataclasses.asdict(obj)
            return {next((o for o, n in _RENAME.items() if n == k), k): v for k, v in d.items()}
        doc: Dict[str, Any] = {"scenario_args": sec(self.scenario), "device_args": sec(self.device),
                               "geo_args": sec(self.geo), "network_args": sec(self.network),
                               "data_args": {"dataset": self.dataset}, "model_args": {"model": self.model},
                               "training_args": {"epochs": self.epochs}, "aggregator_args": {"algorithm": self.algorithm},
                               "tracking_args": sec(self.tracking)}
        doc.update(dataclasses.asdict(self.tunables))
        for k, v in self.extra.items():
            if isinstance(v, dict) and isinstance(doc.get(k), dict):
                doc[k].update(v)
            else:
                doc[k] = v
        return doc


def validate_scenario(parts: List[Dict[str, Any]]) -> List[ParticipantView]:
    """Checks that span the participants of one scenario (the controller's own, controller.py:277-295, plus what the
    runtime assumes): one start node, unique indices / endpoints, one federation, neighbours that exist, CFL = one server."""
    views = [ParticipantView.from_dict(p) for p in parts]
    if not views:
        raise ValueError("a scenario needs at least one participant")
    if sum(v.device.start for v in views) != 1:
        raise ValueError("exactly one participant must be the start node")
    if len({v.scenario.federation for v in views}) != 1:
        raise ValueError("participants disagree on the federation")
    idx = [v.device.idx for v in views]
    if None not in idx and sorted(idx) != list(range(len(views))):
        raise ValueError(f"device_args.idx must enumerate the participants, got {idx}")
    names = [f"{v.network.ip}:{v.network.port}" for v in views]
    if len(set(names)) != len(names):
        raise ValueError("two participants share an ip:port")
    for v, me in zip(views, names):
        for nb in v.network.neighbor_list():
            if nb == me:
                raise ValueError(f"{me} lists itself as a neighbour")
    if views[0].scenario.federation == "CFL" and sum(v.device.role == "server" for v in views) != 1:
        raise ValueError("a CFL scenario has exactly one server")
    return views
