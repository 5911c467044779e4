"""Control-message gossiper (reference: fedstellar/gossiper.py:31-118).

Re-floods hashed control messages (BEAT / ROLE / START_LEARNING / ...) to the neighbours that have
not seen them yet: every ``1 / GOSSIP_MESSAGES_FREC`` seconds at most ``GOSSIP_MESSAGES_PER_ROUND``
(message, peer-set) pairs leave through ``GOSSIP_BROADCAST_EVENT (msg, exclude_list)``.
The thread sleeps on a condition variable while the queue is empty instead of spinning at 100 Hz.
"""
from __future__ import annotations

import threading
import time
from collections import deque

from .utils.observer import Events, Observable


class Gossiper(threading.Thread, Observable):
    def __init__(self, node_name, neighbors, config):
        Observable.__init__(self)
        threading.Thread.__init__(self, name="gossiper-" + node_name, daemon=True)
        self._neighbors = neighbors          # callable returning the current neighbour list, or a list
        self._config = config
        self._pending = deque()              # (msg, [nodes that already have it])
        self._cv = threading.Condition()
        self._terminate = threading.Event()

    def _current_neighbors(self):
        n = self._neighbors
        return list(n() if callable(n) else n)

    def add_messages(self, msgs, node):
        """Queue ``msgs`` (iterable of encoded commands) received from ``node`` for re-flooding."""
        with self._cv:
            for m in msgs:
                self._pending.append((m, [node]))
            self._cv.notify()

    def pending(self) -> int:
        with self._cv:
            return len(self._pending)

    def run(self):
        period = 1.0 / float(self._config.participant["GOSSIP_MESSAGES_FREC"])
        budget_per_round = int(self._config.participant["GOSSIP_MESSAGES_PER_ROUND"])
        while not self._terminate.is_set():
            begin = time.time()
            with self._cv:
                if not self._pending:
                    self._cv.wait(timeout=0.5)
                batch = []
                budget = budget_per_round
                while self._pending and budget > 0:
                    batch.append(self._pending.popleft())
                    budget -= 1
            if batch:
                neighbors = self._current_neighbors()
                for msg, have in batch:
                    # exclude the peers the message came from; the broadcast skips them
                    names = {h if isinstance(h, str) else h.get_name() for h in have}
                    excl = [nc for nc in neighbors if nc.get_name() in names]
                    self.notify(Events.GOSSIP_BROADCAST_EVENT, (msg, excl))
            sleep = period - (time.time() - begin)
            if sleep > 0 and not self._terminate.is_set():
                time.sleep(sleep)

    def stop(self):
        self._terminate.set()
        with self._cv:
            self._cv.notify_all()
