"""Synchronous observer/event bus (reference: fedstellar/utils/observer.py:18-155).

Same event names and ``Observable.notify`` semantics (observers are called inline on the
notifying thread); unlike the reference, a failing observer does not starve the others.
"""
from __future__ import annotations

import logging
import threading


class Events:
    """Event identifiers exchanged between node components."""
    SEND_BEAT_EVENT = "SEND_BEAT_EVENT"
    SEND_ROLE_EVENT = "SEND_ROLE_EVENT"
    END_CONNECTION_EVENT = "END_CONNECTION_EVENT"
    AGGREGATION_FINISHED_EVENT = "AGGREGATION_FINISHED_EVENT"
    CONN_TO_EVENT = "CONN_TO_EVENT"
    START_LEARNING_EVENT = "START_LEARNING_EVENT"
    STOP_LEARNING_EVENT = "STOP_LEARNING_EVENT"
    PARAMS_RECEIVED_EVENT = "PARAMS_RECEIVED_EVENT"
    METRICS_RECEIVED_EVENT = "METRICS_RECEIVED_EVENT"
    TRAIN_SET_VOTE_RECEIVED_EVENT = "TRAIN_SET_VOTE_RECEIVED_EVENT"
    NODE_CONNECTED_EVENT = "NODE_CONNECTED_EVENT"
    PROCESSED_MESSAGES_EVENT = "PROCESSED_MESSAGES_EVENT"
    GOSSIP_BROADCAST_EVENT = "GOSSIP_BROADCAST_EVENT"
    BEAT_RECEIVED_EVENT = "BEAT_RECEIVED_EVENT"
    ROLE_RECEIVED_EVENT = "ROLE_RECEIVED_EVENT"
    REPORT_STATUS_TO_CONTROLLER_EVENT = "REPORT_STATUS_TO_CONTROLLER_EVENT"
    STORE_MODEL_PARAMETERS_EVENT = "STORE_MODEL_PARAMETERS_EVENT"
    # additions of this runtime
    MODELS_READY_EVENT = "MODELS_READY_EVENT"
    TRANSFER_LEADERSHIP_EVENT = "TRANSFER_LEADERSHIP_EVENT"


class Observer:
    """Interface: anything that wants to be told about :class:`Events`."""

    def update(self, event, obj):
        pass


class Observable:
    """Keeps a list of observers and notifies them synchronously."""

    def __init__(self):
        self._observers = []
        self._obs_lock = threading.RLock()

    def add_observer(self, observer):
        with self._obs_lock:
            if observer not in self._observers:
                self._observers.append(observer)

    def remove_observer(self, observer):
        with self._obs_lock:
            if observer in self._observers:
                self._observers.remove(observer)

    def get_observers(self):
        with self._obs_lock:
            return list(self._observers)

    def notify(self, event, obj):
        for o in self.get_observers():
            try:
                o.update(event, obj)
            except Exception:  # one broken observer must not drop the event for the others
                logging.exception("[Observable] observer %r failed on %s", o, event)
