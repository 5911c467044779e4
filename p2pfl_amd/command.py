"""Command pattern: one object per protocol header, bound to a NodeConnection
(reference: fedstellar/command.py:29-167; registered at node_connection.py:95-112)."""
from __future__ import annotations


class Command:
    """Base command; ``execute`` receives the parsed arguments of the message."""

    def __init__(self, node_connection):
        self._node_connection = node_connection

    def execute(self, *args):
        raise NotImplementedError


class Beat_cmd(Command):
    def execute(self, node):
        self._node_connection.notify_heartbeat(node)


class Role_cmd(Command):
    def execute(self, node, role):
        self._node_connection.notify_role(node, role)


class Stop_cmd(Command):
    def execute(self):
        self._node_connection.stop(local=True)


class Conn_to_cmd(Command):
    def execute(self, h, p):
        self._node_connection.notify_conn_to(h, p)


class Start_learning_cmd(Command):
    def execute(self, rounds, epochs):
        self._node_connection.notify_start_learning(rounds, epochs)


class Stop_learning_cmd(Command):
    def execute(self):
        # the reference calls notify_stop_learning() without its required argument
        # (command.py:99-100 vs node_connection.py:428) and drops the link on the TypeError
        self._node_connection.notify_stop_learning(None)


class Params_cmd(Command):
    def execute(self, msg, done):
        self._node_connection.add_param_segment(msg)
        if done:
            params = self._node_connection.get_params()
            self._node_connection.clear_buffer()
            self._node_connection.notify_params(params)


class Models_Ready_cmd(Command):
    def execute(self, round):
        self._node_connection.set_model_ready_status(round)
        self._node_connection.notify_models_ready(round)


class Metrics_cmd(Command):
    def execute(self, node, round, loss, metric):
        self._node_connection.notify_metrics(node, round, loss, metric)


class Vote_train_set_cmd(Command):
    def execute(self, node, votes):
        self._node_connection.notify_train_set_votes(node, votes)


class Models_aggregated_cmd(Command):
    def execute(self, node_list):
        self._node_connection.add_models_aggregated(node_list)


class Model_initialized_cmd(Command):
    def execute(self):
        self._node_connection.set_model_initialized(True)


class Transfer_leadership_cmd(Command):
    def execute(self):
        self._node_connection.set_transfer_leadership(True)
