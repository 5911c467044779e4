"""Heartbeater: liveness beacons + network-wide membership table
(reference: fedstellar/heartbeater.py:32-155).

Every ``HEARTBEAT_PERIOD`` s it emits ``SEND_BEAT_EVENT``; every second beat additionally
``SEND_ROLE_EVENT`` and ``REPORT_STATUS_TO_CONTROLLER_EVENT``; between beats it sweeps the
membership table ``HEARTBEATER_REFRESH_NEIGHBORS_BY_PERIOD`` times and evicts nodes whose last beat
is older than ``NODE_TIMEOUT``. The table holds every node heard of through (gossiped) beats, not
only direct neighbours, and is what ``Node`` uses to validate its train set (node.py:537-548).
The clock is injectable so eviction can be unit-tested without sleeping.
"""
from __future__ import annotations

import threading
import time

from .utils.observer import Events, Observable


class Heartbeater(threading.Thread, Observable):
    def __init__(self, node_name, neighbors, config, clock=time.time):
        Observable.__init__(self)
        threading.Thread.__init__(self, name="heartbeater-" + node_name, daemon=True)
        self._node_name = node_name
        self._neighbors = neighbors
        self._config = config
        self._clock = clock
        self._terminate = threading.Event()
        self._lock = threading.Lock()
        self._nodes = {}        # name -> last beat time
        self._nodes_role = {}   # name -> role
        self._count = 0

    # ------------------------------------------------------------------ thread
    def run(self):
        period = float(self._config.participant["HEARTBEAT_PERIOD"])
        sweeps = max(1, int(self._config.participant["HEARTBEATER_REFRESH_NEIGHBORS_BY_PERIOD"]))
        while not self._terminate.is_set():
            self.notify(Events.SEND_BEAT_EVENT, None)
            self._count += 1
            if self._count % 2 == 0:
                self.notify(Events.SEND_ROLE_EVENT, None)
                self.update_config_with_neighbors()
                self.notify(Events.REPORT_STATUS_TO_CONTROLLER_EVENT, None)
            for _ in range(sweeps):
                self.clear_nodes()
                if self._terminate.wait(period / sweeps):
                    return

    def stop(self):
        self._terminate.set()

    # ------------------------------------------------------------------ membership
    def clear_nodes(self):
        """Evict members not heard of for NODE_TIMEOUT seconds."""
        timeout = float(self._config.participant["NODE_TIMEOUT"])
        now = self._clock()
        with self._lock:
            dead = [n for n, t in self._nodes.items() if now - t > timeout]
            for n in dead:
                self._nodes.pop(n, None)
                self._nodes_role.pop(n, None)   # the reference pops unguarded (heartbeater.py:100-101)
        return dead

    def add_node(self, node):
        """A beat of ``node`` arrived (directly or through gossip)."""
        if node != self._node_name:
            with self._lock:
                self._nodes[node] = self._clock()

    def add_node_role(self, node, role):
        if node != self._node_name:
            with self._lock:
                self._nodes_role[node] = role

    def get_nodes(self):
        """Names of every live node of the network, including this one."""
        with self._lock:
            names = list(self._nodes.keys())
        if self._node_name not in names:
            names.append(self._node_name)
        return names

    def get_nodes_role(self):
        with self._lock:
            return dict(self._nodes_role)

    def update_config_with_neighbors(self):
        """Write the live neighbour string back into the participant config so the controller UI
        shows the current topology (reference heartbeater.py:146-155)."""
        n = self._neighbors
        neighbors = list(n() if callable(n) else n)
        self._config.participant["network_args"]["neighbors"] = " ".join(nc.get_name() for nc in neighbors)
