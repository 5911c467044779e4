"""Aggregator: collects the models of a round and fires the aggregation
(reference: fedstellar/learning/aggregators/aggregator.py:14-238).

Semantics kept:
* ``add_model(model, nodes, weight)`` stores a (partial) aggregate under its contributor set, rejects
  it when a contributor is already present, returns the flattened list of contributors held (the
  ``MODELS_AGGREGATED`` broadcast) or None; the first model of a round arms the
  ``AGGREGATION_TIMEOUT`` (aggregator.py:106-179);
* in *waiting* mode a received model is adopted immediately (``AGGREGATION_FINISHED_EVENT``,
  aggregator.py:119-123) - restricted here to *complete* models (a diffused final model, or an aggregate
  whose contributors cover this node's train set): the reference adopts anything, so an SDFL trainer
  could end a round with another trainer's raw model. Partial aggregates that reach a waiting node are
  kept like on a non-waiting one (that is also what a PROXY relays);
* ``get_partial_aggregation(except_nodes)`` aggregates the stored entries disjoint from what a peer
  already holds -> ``(model, contributors, summed weight)`` (aggregator.py:181-208);
* aggregation runs when the contributors cover the train set, on ``force``, or at the timeout with
  whatever arrived; an empty train set yields ``AGGREGATION_FINISHED_EVENT(None)``
  (aggregator.py:56-61,210-229).

Fixes: the reference re-``__init__``s a started ``Thread`` in ``clear()`` (aggregator.py:231-238) and
lets a stale timeout thread fire into the next round; here every round has a generation number
and a fresh waiter, and ``wait_aggregation()`` lets the node block until the result was delivered
before it advances the round (SURVEY 2.5, round-boundary race).
"""
from __future__ import annotations

import logging
import threading
from typing import Dict, List, Optional, Tuple

from ...role import Role
from ...utils.observer import Events, Observable


class Aggregator(Observable):
    def __init__(self, node_name="unknown", config=None):
        Observable.__init__(self)
        self.node_name = node_name
        self.config = config
        self._lock = threading.RLock()
        self._generation = 0
        self._reset_state()

    # kept for API compatibility with the thread-based reference class
    daemon = True

    def _reset_state(self):
        self._train_set: List[str] = []
        self._waiting_aggregated_model = False
        self._models: Dict[str, Tuple[object, float]] = {}
        self._ready = threading.Event()       # contributors cover the train set (or forced)
        self._done = threading.Event()        # AGGREGATION_FINISHED delivered
        self._armed = False

    @property
    def role(self):
        return self.config.participant["device_args"]["role"] if self.config is not None else Role.AGGREGATOR

    def _timeout(self) -> float:
        return float(self.config.participant["AGGREGATION_TIMEOUT"]) if self.config is not None else 60.0

    # ------------------------------------------------------------------ to be provided by subclasses
    def aggregate(self, models):
        """models: {"nodeA nodeB": (state_dict, weight)} -> aggregated state_dict."""
        raise NotImplementedError

    # ------------------------------------------------------------------ configuration
    def set_nodes_to_aggregate(self, listnodes):
        with self._lock:
            self._train_set = list(listnodes)

    def get_nodes_to_aggregate(self):
        with self._lock:
            return list(self._train_set)

    def set_waiting_aggregated_model(self):
        logging.info("[Aggregator] set_waiting_aggregated_model = True")
        with self._lock:
            self._waiting_aggregated_model = True

    def get_waiting_aggregated_model(self):
        return self._waiting_aggregated_model

    def get_models_added(self) -> List[str]:
        with self._lock:
            return [n for key in self._models for n in key.split()]

    # ------------------------------------------------------------------ round machinery
    def _arm(self):
        """Start this round's waiter (first model of the round)."""
        self._armed = True
        gen = self._generation
        ready, done = self._ready, self._done
        t = threading.Thread(target=self._wait_and_aggregate, args=(gen, ready, done),
                             name=f"aggregator-{self.node_name}-{gen}", daemon=True)
        t.start()

    def _wait_and_aggregate(self, gen, ready, done):
        ready.wait(timeout=self._timeout())
        with self._lock:
            if gen != self._generation:
                return                         # the round was cleared meanwhile
            train_set = list(self._train_set)
            models = dict(self._models)
        if not train_set:
            logging.info("[Aggregator] empty train set -> no aggregation")
            self.notify(Events.AGGREGATION_FINISHED_EVENT, None)
            done.set()
            return
        have = [n for key in models for n in key.split()]
        if len(have) != len(train_set):
            logging.info("[Aggregator] aggregating with missing models: %s", set(train_set) - set(have))
        result = self.aggregate(models) if models else None
        with self._lock:
            # delivered under the lock: clear() (round advance) waits for a delivery in progress and a
            # result of a cleared round is never handed to the next one
            if gen != self._generation:
                return
            self.notify(Events.AGGREGATION_FINISHED_EVENT, result)
            done.set()

    def run(self):
        """The reference's thread body (aggregator.py:40-74: wait for every model of the train set or for
        AGGREGATION_TIMEOUT, aggregate what is there, notify AGGREGATION_FINISHED). Here a waiter thread is armed by
        the first model of a round; calling ``run()`` does the same wait synchronously for the current round - for
        code that drove the reference class as ``aggregator.run()`` / a started thread."""
        with self._lock:
            if self._armed:                    # this round already has its waiter: just wait for its delivery
                done = self._done
                gen = None
            else:
                self._armed = True
                gen, ready, done = self._generation, self._ready, self._done
        if gen is None:
            done.wait(self._timeout() + 5.0)
        else:
            self._wait_and_aggregate(gen, ready, done)

    def wait_aggregation(self, timeout: Optional[float] = None) -> bool:
        """Block until this round's result was delivered (True) or ``timeout`` elapsed."""
        return self._done.wait(timeout if timeout is not None else self._timeout() + 5.0)

    def add_model(self, model, nodes, weight, check=True):
        """``check=False`` defers the completeness check: the caller announces the new contributor set
        (MODELS_AGGREGATED) first and then calls :meth:`check_and_run_aggregation` itself, so that a
        peer never sees this round's MODELS_READY before the last MODELS_AGGREGATED."""
        if self._waiting_aggregated_model and self.is_complete(nodes):
            logging.info("[Aggregator] received a model from %s while waiting -> adopting it", nodes)
            self.notify(Events.AGGREGATION_FINISHED_EVENT, model)
            self._done.set()
            return None
        if nodes is None:
            logging.debug("[Aggregator] model received by diffusion while not waiting: ignored")
            return None
        with self._lock:
            if self._train_set and not self._armed:
                self._arm()
            models_added = [n for key in self._models for n in key.split()]
            if len(self._train_set) > len(models_added):
                if all(n not in models_added for n in nodes):
                    self._models[" ".join(nodes)] = (model, weight)
                    logging.info("[Aggregator] Model added (%d/%d) from %s", len(models_added) + len(nodes),
                                 len(self._train_set), nodes)
                    if check:
                        self.check_and_run_aggregation()
                    return models_added + list(nodes)
                logging.debug("[Aggregator] can't add a model that has already been added %s", nodes)
        return None

    def get_single(self, name):
        """(model, [name], weight) of the un-aggregated contribution of ``name`` if it is held, else None."""
        with self._lock:
            entry = self._models.get(name)
        return None if entry is None else (entry[0], [name], entry[1])

    def is_complete(self, nodes) -> bool:
        """A model a waiting node may adopt as the round's result: diffused (no contributor list) or
        aggregated from at least this node's whole train set (an empty train set accepts anything:
        IDLE / PROXY nodes)."""
        if nodes is None:
            return True
        with self._lock:
            return set(self._train_set) <= set(nodes)

    def get_partial_aggregation(self, except_nodes):
        with self._lock:
            models = dict(self._models)
        chosen, nodes_aggregated, weight = {}, [], 0
        for key, (m, w) in models.items():
            names = key.split()
            if all(n not in except_nodes for n in names):
                chosen[key] = (m, w)
                nodes_aggregated += names
                weight += w
        if not chosen:
            return None, None, None
        return self.aggregate(chosen), nodes_aggregated, weight

    def check_and_run_aggregation(self, force=False):
        with self._lock:
            have = [n for key in self._models for n in key.split()]
            if (force or len(have) >= len(self._train_set)) and self._train_set:
                self._ready.set()

    def clear(self):
        """Forget the round (observers are preserved)."""
        with self._lock:
            self._generation += 1
            self._ready.set()                  # release a waiter of the old generation (it will exit)
            self._reset_state()
