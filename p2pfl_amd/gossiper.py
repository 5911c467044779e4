"""Control-message gossiper (reference: fedstellar/gossiper.py:31-118).

Re-floods hashed control messages (BEAT / ROLE / START_LEARNING / ...) to the neighbours that have
not seen them yet: every ``1 / GOSSIP_MESSAGES_FREC`` seconds at most ``GOSSIP_MESSAGES_PER_ROUND``
sends - (message, peer) pairs, as in the reference - leave through ``GOSSIP_BROADCAST_EVENT (msg, exclude_list)``;
a message that does not fit completely is delivered to the peers that do and keeps the rest for the next tick.
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
                pending = list(self._pending)
                self._pending.clear()
            # the budget counts SENDS, i.e. (message, peer) pairs (gossiper.py:84-103): a message whose remaining
            # peers do not fit goes to as many as do and waits for the next tick with the rest
            budget = budget_per_round
            neighbors = self._current_neighbors() if pending else []
            keep = []
            for k, (msg, have) in enumerate(pending):
                if budget <= 0:
                    keep.extend(pending[k:])
                    break
                names = {h if isinstance(h, str) else h.get_name() for h in have}
                todo = [nc for nc in neighbors if nc.get_name() not in names]
                excl = [nc for nc in neighbors if nc.get_name() in names]
                if len(todo) > budget:
                    later = todo[budget:]
                    self.notify(Events.GOSSIP_BROADCAST_EVENT, (msg, excl + later))
                    keep.append((msg, list(have) + [nc.get_name() for nc in todo[:budget]]))
                    budget = 0
                else:
                    self.notify(Events.GOSSIP_BROADCAST_EVENT, (msg, excl))
                    budget -= len(todo)
            if keep:
                with self._cv:
                    self._pending.extendleft(reversed(keep))
            sleep = period - (time.time() - begin)
            if sleep > 0 and not self._terminate.is_set():
                time.sleep(sleep)

    def stop(self):
        self._terminate.set()
        with self._cv:
            self._cv.notify_all()
