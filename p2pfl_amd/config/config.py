"""Scenario / participant configuration (reference: fedstellar/config/config.py:20-109).

The real configuration surface is the per-participant JSON (same keys and sections as
``fedstellar/config/participant.json.example:1-83``); code all over the runtime reads it as a
plain mutable dict (``config.participant["GOSSIP_MODELS_FREC"]``) and the heartbeater / leadership
transfer write into it, so that contract is kept. On top of it this build adds defaults for every
key (the reference's ``_set_default_config`` is a stub), light validation, and two optional
``device_args`` extensions (``gpu``, ``placement``) used by the one-process-per-GPU launcher.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional

AES_BLOCK = 16
_EXAMPLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "participant.json.example")

TUNABLES = ("BLOCK_SIZE", "NODE_TIMEOUT", "VOTE_TIMEOUT", "AGGREGATION_TIMEOUT", "HEARTBEAT_PERIOD",
            "HEARTBEATER_REFRESH_NEIGHBORS_BY_PERIOD", "WAIT_HEARTBEATS_CONVERGENCE", "TRAIN_SET_SIZE",
            "TRAIN_SET_CONNECT_TIMEOUT", "AMOUNT_LAST_MESSAGES_SAVED", "GOSSIP_MESSAGES_FREC",
            "GOSSIP_MESSAGES_PER_ROUND", "GOSSIP_EXIT_ON_X_EQUAL_ROUNDS", "GOSSIP_MODELS_FREC",
            "GOSSIP_MODELS_PER_ROUND")


def default_participant() -> Dict[str, Any]:
    """A fresh copy of the default participant dict (core defaults of the reference)."""
    with open(_EXAMPLE) as f:
        return json.load(f)


def _deep_update(base: Dict[str, Any], override: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _deep_update(base[k], v)
        else:
            base[k] = v
    return base


def fast_timings(participant: Dict[str, Any], scale: float = 0.05) -> Dict[str, Any]:
    """Shrink the protocol's pacing constants (tests / co-resident nodes do not need WAN timings)."""
    p = participant
    p["HEARTBEAT_PERIOD"] = max(0.05, p["HEARTBEAT_PERIOD"] * scale)
    p["WAIT_HEARTBEATS_CONVERGENCE"] = max(0.1, p["WAIT_HEARTBEATS_CONVERGENCE"] * scale)
    p["GOSSIP_MODELS_FREC"] = p["GOSSIP_MODELS_FREC"] / scale
    p["NODE_TIMEOUT"] = max(2.0, p["NODE_TIMEOUT"] * scale * 4)
    p["AGGREGATION_TIMEOUT"] = max(2.0, p["AGGREGATION_TIMEOUT"] * scale * 2)
    return p


class Config:
    """Holds ``participant`` (node side) or ``participants`` (controller side).

    Args:
        entity: "participant" or "controller" (kept for API compatibility).
        topology_config_file: optional JSON with the topology (controller side).
        participant_config_file: participant JSON; missing keys fall back to defaults.
    """

    topology: Dict[str, Any] = {}
    participant: Dict[str, Any] = {}
    participants: List[Dict[str, Any]] = []
    participants_path: List[str] = []

    def __init__(self, entity="participant", topology_config_file=None, participant_config_file=None,
                 participant: Optional[Dict[str, Any]] = None):
        self.entity = entity
        self.topology = {}
        self.participant = {}
        self.participants = []
        self.participants_path = []
        if topology_config_file is not None:
            self.set_topology_config(topology_config_file)
        if participant_config_file is not None:
            self.set_participant_config(participant_config_file)
        elif participant is not None:
            self.participant = _deep_update(default_participant(), participant)
            self._adjust_block_size()

    def __getstate__(self):
        return {"topology": self.topology, "participant": self.participant}

    def __setstate__(self, d):
        self.topology = d["topology"]
        self.participant = d["participant"]
        self.entity = "participant"
        self.participants, self.participants_path = [], []

    def get_topology_config(self):
        return json.dumps(self.topology, indent=2)

    def get_participant_config(self):
        return json.dumps(self.participant, indent=2)

    def _set_default_config(self):
        self.participant = default_participant()
        self._adjust_block_size()

    def set_topology_config(self, topology_config_file):
        with open(topology_config_file) as f:
            self.topology = json.load(f)

    def set_participant_config(self, participant_config_file):
        with open(participant_config_file) as f:
            loaded = json.load(f)
        self.participant = _deep_update(default_participant(), loaded)
        self.validate(self.participant)
        self._adjust_block_size()

    def set_participants_config(self, participants_config):
        self.participants, self.participants_path = [], []
        for path in participants_config:
            with open(path) as f:
                self.participants.append(_deep_update(default_participant(), json.load(f)))
            self.participants_path.append(path)

    def add_participant_config(self, participant_config_file):
        with open(participant_config_file) as f:
            self.participants.append(_deep_update(default_participant(), json.load(f)))
        self.participants_path.append(participant_config_file)

    # ------------------------------------------------------------------
    def _adjust_block_size(self):
        """Round BLOCK_SIZE up to a multiple of the AES block (reference config.py:86-109): an
        encrypted fragment must stay exactly one BLOCK_SIZE read on the receiver."""
        bs = int(self.participant.get("BLOCK_SIZE", 2048))
        rest = bs % AES_BLOCK
        if rest != 0:
            bs += AES_BLOCK - rest
        self.participant["BLOCK_SIZE"] = bs

    # kept under the reference's mangled-looking public alias too
    _Config__adjust_block_size = _adjust_block_size

    def view(self):
        """Typed, validated snapshot of ``participant`` (config/schema.py); the dict stays the live, mutable state."""
        from .schema import ParticipantView
        return ParticipantView.from_dict(self.participant)

    @staticmethod
    def validate(p: Dict[str, Any]) -> None:
        from .schema import ParticipantView
        ParticipantView.from_dict(p)          # types, ranges, known federations / roles / accelerators
        for key in TUNABLES:
            if key not in p:
                raise ValueError(f"participant config misses {key}")
            if not isinstance(p[key], (int, float)) or p[key] <= 0:
                raise ValueError(f"participant config: {key} must be a positive number, got {p[key]!r}")
        fed = p["scenario_args"].get("federation", "DFL")
        if fed not in ("DFL", "SDFL", "CFL"):
            raise ValueError(f"unknown federation {fed!r}")
        role = p["device_args"].get("role")
        if role not in ("trainer", "aggregator", "proxy", "idle", "server"):
            raise ValueError(f"unknown role {role!r}")
