"""Virtual membership + deterministic fault injection for nodes that share a process.

The reference keeps, per node, a heartbeater thread that floods ``BEAT`` every ``HEARTBEAT_PERIOD`` seconds
and evicts peers whose last beat is older than ``NODE_TIMEOUT`` (fedstellar/heartbeater.py:66-111), plus one
socket per neighbour whose receive timeout tears the link down (node_connection.py:150,225-242). Hundreds
of co-resident virtual nodes do not need hundreds of threads and sockets to agree that one of them is gone:
this module keeps ONE table ``last_beat[node]`` on a virtual clock that the round loop advances, with the
same two constants and the same observable behaviour:

* a node that is killed / paused stops beating at once (``running()``), but its neighbours go on BELIEVING in
  it (``believed()``) until ``NODE_TIMEOUT`` has passed on the clock without a beat: until then it stays in
  their train sets, they wait for its model, give up after ``AGGREGATION_TIMEOUT`` and aggregate the subset
  that did arrive (aggregator.py:47-76) - the round loop charges that wait to the clock, which is what finally
  evicts the node (heartbeater.py:88-101) and shrinks the train sets;
* faults land at the start of a round or in the middle of it (``phase="mid"``: the node trained, its model never
  arrives);
* a revived / resumed node beats again and is a member from its next beat on;
* ``drop_link`` / ``restore_link`` remove and restore single edges of the topology (socket errors).

Faults are scheduled per round (``schedule(round, action, node[, peer])``), so churn scenarios
(BASELINE config #4) are reproducible.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

ACTIONS = ("kill", "revive", "pause", "resume", "drop_link", "restore_link")


@dataclass
class Fault:
    action: str
    node: int
    peer: Optional[int] = None
    phase: str = "start"          # "start": before evaluation / training; "mid": after training, before the exchange


class VirtualMembership:
    def __init__(self, n_nodes: int, heartbeat_period: float = 4.0, node_timeout: float = 20.0,
                 instant: bool = True):
        """``instant``: evict at the fault itself (the fixed point of detection) instead of NODE_TIMEOUT later."""
        self.n = n_nodes
        self.period = float(heartbeat_period)
        self.timeout = float(node_timeout)
        self.instant = instant
        self.clock = 0.0
        self.beating = np.ones(n_nodes, dtype=bool)          # the node's own heartbeater runs
        self.last_beat = np.zeros(n_nodes, dtype=np.float64)
        self.member = np.ones(n_nodes, dtype=bool)            # what every other node believes
        self.dropped: set = set()                             # {(i, j)} with i < j
        self.schedule_table: Dict[int, List[Fault]] = {}
        self.events: List[Tuple[float, str, int]] = []        # (clock, what, node) log for observability

    # ------------------------------------------------------------------ faults
    def schedule(self, round_idx: int, action: str, node: int, peer: Optional[int] = None,
                 phase: str = "start") -> None:
        if action not in ACTIONS:
            raise ValueError(f"unknown fault action {action!r}; known: {ACTIONS}")
        if action.endswith("link") and peer is None:
            raise ValueError(f"{action} needs a peer")
        if phase not in ("start", "mid"):
            raise ValueError(f"unknown fault phase {phase!r}")
        self.schedule_table.setdefault(int(round_idx), []).append(
            Fault(action, int(node), None if peer is None else int(peer), phase))

    def apply_round(self, round_idx: int, phase: Optional[str] = "start") -> List[Fault]:
        """Apply the faults scheduled for ``round_idx`` and ``phase`` (None: every phase - replaying a finished round)."""
        done = [f for f in self.schedule_table.get(round_idx, []) if phase is None or f.phase == phase]
        for f in done:
            if f.action in ("kill", "pause"):
                self.beating[f.node] = False
                self.events.append((self.clock, f.action, f.node))
                if self.instant:
                    self.member[f.node] = False
            elif f.action in ("revive", "resume"):
                self.beating[f.node] = True
                self.last_beat[f.node] = self.clock
                self.member[f.node] = True
                self.events.append((self.clock, f.action, f.node))
            elif f.action == "drop_link":
                self.dropped.add((min(f.node, f.peer), max(f.node, f.peer)))
            elif f.action == "restore_link":
                self.dropped.discard((min(f.node, f.peer), max(f.node, f.peer)))
        return done

    # ------------------------------------------------------------------ clock
    def advance(self, seconds: float) -> List[int]:
        """Advance the virtual clock: live nodes beat every period, silent ones are evicted after the
        timeout. Returns the nodes evicted by this call."""
        self.clock += max(float(seconds), 0.0)
        beats = np.floor(self.clock / self.period) * self.period
        self.last_beat[self.beating] = np.maximum(self.last_beat[self.beating], beats)
        evicted = np.nonzero(self.member & ~self.beating & (self.clock - self.last_beat > self.timeout))[0]
        for i in evicted:
            self.member[i] = False
            self.events.append((self.clock, "evicted", int(i)))
        return [int(i) for i in evicted]

    def note(self, what: str, node: int) -> None:
        """Log an observation of the round loop (e.g. ``aggregation_timeout``) on the membership clock."""
        self.events.append((self.clock, what, int(node)))

    # ------------------------------------------------------------------ views
    def running(self) -> np.ndarray:
        """Nodes whose own process is up (they evaluate, train, contribute)."""
        return self.beating

    def believed(self) -> np.ndarray:
        """Nodes the rest of the network still counts as members (train-set membership)."""
        return self.member

    def alive(self) -> np.ndarray:
        """Nodes that are up AND counted in: beating and believed."""
        return self.member & self.beating

    def get_nodes(self) -> List[int]:
        """Network-wide membership as every node sees it (``BaseNode.get_network_nodes``)."""
        return [int(i) for i in np.nonzero(self.member)[0]]

    def link_up(self, i: int, j: int) -> bool:
        return (min(i, j), max(i, j)) not in self.dropped
