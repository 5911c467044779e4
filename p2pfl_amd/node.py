"""Node: the federated-learning state machine on top of BaseNode
(reference: fedstellar/node.py:51-1017).

Public surface kept: ``Node(idx, experiment_name, model, data, hostdemo, host, port, config, learner,
encrypt)``, ``start/stop/connect_to/set_start_learning/set_stop_learning/set_data/set_model/add_model/
update``, attributes ``round`` (None <=> not learning), ``totalrounds``, ``learner``, ``aggregator``.

Round logic per role follows node.py:402-535 (aggregator / server / trainer / proxy / idle), model
exchange is the push gossip of node.py:692-809 over ``PARAMS`` frames, SDFL hands the aggregator
role to a random neighbour after each round (node.py:676-686).

Structural changes: one-shot gates are ``threading.Event``s instead of pre-acquired
locks; received payloads are decoded/aggregated on a worker thread, not on the socket receive
thread; the round loop is iterative (the reference recurses ``__train_step`` ->
``__on_round_finished`` -> ``__train_step``); an aggregating node waits until its aggregation was
delivered before it clears the aggregator and advances the round; a waiting node adopts an
aggregate that arrives early instead of sleeping a fixed 5 s (node.py:490).
"""
from __future__ import annotations

import json
import logging
import os
import queue
import random
import threading
import time
from datetime import datetime, timedelta

from .base_node import BaseNode
from .communication_protocol import CommunicationProtocol
from .config import Config
from .learning.aggregators.fedavg import FedAvg
from .learning.codec import load_checkpoint as _load_ckpt
from .learning.codec import save_checkpoint as _save_ckpt
from .learning.exceptions import DecodingParamsError, ModelNotMatchingError
from .learning.pytorch.lightninglearner import LightningLearner
from .learning.pytorch.remotelogger import FedstellarWBLogger
from .learning.pytorch.statisticslogger import CSVLogger, FedstellarLogger
from .role import Role
from .utils.observer import Events


class Node(BaseNode):
    def __init__(self, idx, experiment_name, model, data, hostdemo=None, host="127.0.0.1", port=None,
                 config=Config, learner=LightningLearner, encrypt=False):
        if config is None or isinstance(config, type):
            config = Config(participant={})
        BaseNode.__init__(self, experiment_name, hostdemo, host, port, encrypt, config)
        self.idx = idx
        self.config = config
        self._report_status_to_controller()

        # learning state
        self.round = None
        self.totalrounds = None
        self._model_initialized = False
        self._initial_neighbors = []
        self._start_thread_lock = threading.Lock()
        self._learning_thread = None

        # learner + logger (reference node.py:112-129)
        tr = self.config.participant["tracking_args"]
        metrics_dir = self.log_dir
        if tr.get("enable_remote_tracking"):
            logger = FedstellarWBLogger(project=tr.get("wandb_project") or "fedstellar", group=self.experiment_name,
                                        name=f"participant_{self.idx}", save_dir=metrics_dir)
        elif tr.get("local_tracking") == "web":
            logger = FedstellarLogger(metrics_dir, name="metrics", version=f"participant_{self.idx}", log_graph=True)
        elif tr.get("local_tracking") == "csv" and tr.get("log_dir"):
            logger = CSVLogger(metrics_dir, name="metrics", version=f"participant_{self.idx}")
        else:
            logger = None
        self.learner = learner(model, data, config=self.config, logger=logger)
        logging.info("[NODE] Role: %s", self.role)

        algo = self.config.participant["aggregator_args"].get("algorithm", "FedAvg")
        if algo != "FedAvg":
            raise NotImplementedError(f"Aggregation algorithm {algo} not supported")
        self.aggregator = FedAvg(node_name=self.get_name(), config=self.config)
        self.aggregator.add_observer(self)
        self.shared_metrics = False

        self._stored_model_parameters = []
        self._proxy_timeout = datetime.now()
        self._train_set = []
        self._train_set_votes = {}
        self._train_set_votes_lock = threading.Lock()

        # gates
        self._init_model_ready = threading.Event()      # initial weights installed
        self._aggregation_done = threading.Event()      # this round's aggregate was adopted
        self._round_lock = threading.RLock()            # round advance vs. intake of a decoded payload
        self._model_queue: "queue.Queue[bytes]" = queue.Queue()
        self._model_worker = threading.Thread(target=self._model_worker_loop, name="models-" + self.get_name(),
                                              daemon=True)
        self._model_worker.start()
        self._ready_sent = False
        self._local_done = False
        self.checkpoint_dir = tr.get("checkpoint_dir") or None
        self.round_times = []

    # ------------------------------------------------------------------ helpers
    @property
    def role(self):
        return self.config.participant["device_args"]["role"]

    def _p(self, key):
        return self.config.participant[key]

    # ------------------------------------------------------------------ node management
    def connect_to(self, h, p, full=False, force=False):
        if self.round is not None and not force:
            logging.info("[NODE] Cant connect to other nodes when learning is running.")
            return None
        return super().connect_to(h, p, full, force)

    def stop(self):
        if self.round is not None:
            self._stop_learning()
        self._model_queue.put(None)
        self.learner.close()
        super().stop()

    def set_data(self, data):
        self.learner.set_data(data)

    def set_model(self, model):
        self.learner.set_model(model)

    # ------------------------------------------------------------------ checkpoint / resume
    def save_checkpoint(self, path):
        """Write the current model in the payload format (+ JSON sidecar)."""
        sd = self.learner.get_parameters()
        _save_ckpt(path, [v.detach().cpu().numpy() for v in sd.values()], [self.get_name()],
                   self.learner.get_num_samples()[0],
                   meta={"round": self.round, "role": self.role, "node": self.get_name(),
                         "optimizer_policy": "reset-per-round"})

    def load_checkpoint(self, path):
        with open(path, "rb") as f:
            sd, _, _ = self.learner.decode_parameters(f.read())
        if not self.learner.check_parameters(sd):
            raise ModelNotMatchingError("checkpoint does not match the model")
        self.learner.set_parameters(sd)
        _, _, _, meta = _load_ckpt(path)
        return meta

    # ------------------------------------------------------------------ network learning management
    def set_start_learning(self, rounds=1, epochs=1):
        if self._terminate_flag.is_set():
            logging.info("[NODE] Node must be running to start learning")
            return
        if self.round is None:
            logging.info("[NODE] I am the initializer node | START_LEARNING rounds=%s epochs=%s", rounds, epochs)
            self.broadcast(CommunicationProtocol.build_start_learning_msg(rounds, epochs))
            self.broadcast(CommunicationProtocol.build_model_initialized_msg())
            self._model_initialized = True
            self._init_model_ready.set()
            self._start_learning_thread(rounds, epochs)
        else:
            logging.info("[NODE] Learning already started")

    def set_stop_learning(self):
        if self.round is not None:
            self.broadcast(CommunicationProtocol.build_stop_learning_msg())
            self._stop_learning()
        else:
            logging.info("[NODE] Learning already stopped")

    def _start_learning_thread(self, rounds, epochs):
        t = threading.Thread(target=self._start_learning, args=(rounds, epochs),
                             name="learning_thread-" + self.get_name(), daemon=True)
        self._learning_thread = t
        t.start()

    def _start_learning(self, rounds, epochs):
        with self._start_thread_lock:          # avoid duplicated training threads
            if self.round is not None:
                return
            self.round = 0
            self._aggregation_done.clear()
            self.totalrounds = rounds
        self.learner.init()
        begin = time.time()
        self._gossip_model_difusion(initialization=True)
        wait = float(self._p("WAIT_HEARTBEATS_CONVERGENCE")) - (time.time() - begin)
        if wait > 0:
            time.sleep(wait)
        self._initial_neighbors = self.get_neighbors()
        logging.info("[NODE] Learning started in %s | rounds %s | epochs %s", self.get_name(), rounds, epochs)
        self.learner.set_epochs(epochs)
        self.learner.create_trainer()
        while self.round is not None:
            t0 = time.time()
            self._train_step()
            if self.round is None:
                break
            finished = self._on_round_finished()
            self.round_times.append(time.time() - t0)
            if finished:
                break
        logging.info("[NODE] learning thread finished in %s", self.get_name())

    def _stop_learning(self):
        logging.info("[NODE] Stopping learning")
        self.round = None
        self.totalrounds = None
        self.learner.interrupt_fit()
        self.aggregator.check_and_run_aggregation(force=True)
        self.aggregator.set_nodes_to_aggregate([])
        self.aggregator.clear()
        self._aggregation_done.set()
        self._init_model_ready.set()
        self._model_initialized = False

    # ------------------------------------------------------------------ model intake
    def _model_worker_loop(self):
        while True:
            m = self._model_queue.get()
            if m is None:
                return
            try:
                self.add_model(*m) if isinstance(m, tuple) else self.add_model(m)
            except Exception as e:
                logging.exception(e)

    def add_model(self, m, payload_round=None):
        """A complete payload arrived: either the initial model, or a (partial) aggregate.
        ``payload_round``: round the link attributes the payload to (NodeConnection.payload_round);
        a late duplicate of the previous round must not count as a contribution to this one."""
        if self.round is None:
            logging.error("[NODE] Tried to add a model while learning is not running")
            return
        if self._model_initialized and payload_round is not None and payload_round != self.round:
            logging.debug("[NODE] dropped a payload of round %s in round %s", payload_round, self.round)
            return
        try:
            if not self._model_initialized:
                model, _, _ = self.learner.decode_parameters(m)
                if not self.learner.check_parameters(model):
                    raise ModelNotMatchingError("Not matching models")
                self.learner.set_parameters(model)
                self._model_initialized = True
                logging.info("[NODE] Initialization Model Weights")
                self._init_model_ready.set()
                self.broadcast(CommunicationProtocol.build_model_initialized_msg())
                return
            decoded, contributors, weight = self.learner.decode_parameters(m)
            if not self.learner.check_parameters(decoded):
                raise ModelNotMatchingError("Not matching models")
            if contributors is not None and self._train_set and any(c not in self._train_set for c in contributors):
                # FedAvg runs over train_set = neighbours U {self} (node.py:411-416): a partial aggregate that carries a
                # model from outside it (a neighbour's other neighbour, in a ring / random graph) is not ours to average
                logging.debug("[NODE] refused contributors outside the train set: %s", contributors)
                return
            waits = Role.waits(self.role)
            with self._round_lock:     # decoding took milliseconds: the round may have advanced meanwhile
                if payload_round is not None and payload_round != self.round:
                    return
                if self._aggregation_done.is_set():
                    return      # this round's aggregate is already in place: a duplicate must not be re-announced
                if waits and not self._local_done and self.aggregator.is_complete(contributors):
                    # the round's result overtook our own training: it cannot replace the model while the
                    # local fit is running; the sender repeats it until we announce MODELS_READY. Partial
                    # aggregates ARE kept, so a round without a leader (TRANSFER_LEADERSHIP still in
                    # flight) completes by every trainer averaging its train set itself.
                    return
                early_aggregate = (waits and not self.aggregator.get_waiting_aggregated_model()
                                   and self.aggregator.is_complete(contributors))
                if early_aggregate:
                    # the aggregate overtook our own "waiting" transition: adopt it now
                    self.update(Events.AGGREGATION_FINISHED_EVENT, decoded)
                    return
                models_added = self.aggregator.add_model(decoded, contributors, weight, check=False)
            if models_added is not None:
                self.broadcast(CommunicationProtocol.build_models_aggregated_msg(models_added))
                self.aggregator.check_and_run_aggregation()
        except DecodingParamsError as e:
            logging.error("[NODE] Error decoding parameters: %s", e)
            self.stop()
        except ModelNotMatchingError as e:
            logging.error("[NODE] Models not matching: %s", e)
            self.stop()

    # ------------------------------------------------------------------ one round
    def _train_step(self):
        if self.round is None:
            return
        for n in self.get_neighbors():
            if n.get_name() not in self._train_set:
                self._train_set.append(n.get_name())
        if self.get_name() not in self._train_set:
            self._train_set.append(self.get_name())
        self._validate_train_set()
        self._ready_sent = False
        self._local_done = Role.waits(self.role) and not Role.trains(self.role)   # proxy / idle: nothing to do
        role = self._round_role = self.role

        if role in (Role.AGGREGATOR, Role.SERVER):
            self.aggregator.set_nodes_to_aggregate(self._train_set)
            if self.round is not None:
                self._evaluate()
            if self.round is not None and role != Role.SERVER:      # the CFL server never trains
                self._train()
            if self.round is not None:
                # announce first: adding the last missing model aggregates and sends MODELS_READY, and the
                # link-order rule (NodeConnection.payload_round) needs MODELS_AGGREGATED before it
                self.broadcast(CommunicationProtocol.build_models_aggregated_msg([self.get_name()]))
                self.aggregator.add_model(self._snapshot_model(), [self.get_name()],
                                          self.learner.get_num_samples()[0])
                if role == Role.SERVER:
                    self._gossip_model_difusion()
                else:
                    self._gossip_model_aggregation()
                # do not advance before the aggregate was adopted (reference race)
                self._aggregation_done.wait(float(self._p("AGGREGATION_TIMEOUT")) + 5.0)
                if role == Role.AGGREGATOR and self._p("scenario_args").get("federation", "DFL") != "DFL":
                    # SDFL / mixed-role scenarios: neighbours in waiting roles take this node's result
                    # (the reference leaves them with whatever payload arrived first, aggregator.py:119-123)
                    self._gossip_model_difusion()

        elif role == Role.TRAINER:
            self.aggregator.set_nodes_to_aggregate(self._train_set)
            if self.round is not None:
                self._evaluate()
            if self.round is not None:
                self._train()
            if self.round is not None:
                self._local_done = True
                self.broadcast(CommunicationProtocol.build_models_aggregated_msg([self.get_name()]))
                self.aggregator.add_model(self._snapshot_model(), [self.get_name()],
                                          self.learner.get_num_samples()[0])
                self._gossip_model_aggregation()
                if not self._aggregation_done.is_set():
                    self.aggregator.set_waiting_aggregated_model()
                    self._aggregation_done.wait(float(self._p("AGGREGATION_TIMEOUT")))

        elif role == Role.PROXY:
            if self.round is not None:
                self.broadcast(CommunicationProtocol.build_models_aggregated_msg([self.get_name()]))
                if datetime.now() > self._proxy_timeout:
                    self._gossip_model_aggregation()
                    self._proxy_timeout = datetime.now() + timedelta(seconds=10)
                self.aggregator.set_waiting_aggregated_model()
                self._aggregation_done.wait(float(self._p("AGGREGATION_TIMEOUT")))

        elif role == Role.IDLE:
            self.aggregator.set_waiting_aggregated_model()
            self._aggregation_done.wait(float(self._p("AGGREGATION_TIMEOUT")))
        else:
            logging.warning("[NODE] Role %s not implemented", role)

    def _snapshot_model(self):
        """Copy of the trained model for the aggregator. ``state_dict()`` hands out the live tensors and
        ``set_parameters`` overwrites them in place, so without the copy the contribution this node keeps
        (and forwards inside partial aggregates) would silently turn into whatever model it adopts later."""
        return type(self.learner.get_parameters())(
            (k, v.detach().clone()) for k, v in self.learner.get_parameters().items())

    def _validate_train_set(self):
        live = set(self.get_network_nodes())
        self._train_set = [n for n in self._train_set if n in live or n == self.get_name()]
        logging.info("[NODE] Train set of %d nodes: %s", len(self._train_set), self._train_set)

    def _train(self):
        logging.info("[NODE] Start training...")
        self.learner.fit()
        logging.info("[NODE] Finish training...")

    def _evaluate(self):
        logging.info("[NODE] Start evaluation...")
        results = self.learner.evaluate()
        if results is not None and self.shared_metrics:
            self.broadcast(CommunicationProtocol.build_metrics_msg(self.get_name(), self.round, results[0], results[1]))
        logging.info("[NODE] Finish evaluation...")

    def _wait_neighbors_ready(self):
        """Round barrier: MODELS_AGGREGATED carries no round number, so a node that raced ahead would
        mix the bookkeeping of two rounds. Every node announces MODELS_READY <round> once it holds the
        round's model; the next round starts when all live train-set neighbours have done so (or the
        aggregation timeout expires)."""
        if not self._ready_sent:
            self.broadcast(CommunicationProtocol.build_models_ready_msg(self.round))
            self._ready_sent = True
        deadline = time.time() + float(self._p("AGGREGATION_TIMEOUT"))
        rnd = self.round
        while time.time() < deadline and self.round is not None:
            pending = [nc for nc in self.get_neighbors()
                       if nc.get_name() in self._train_set and nc.get_model_ready_status() < rnd]
            if not pending:
                return
            time.sleep(0.01)

    def _on_round_finished(self) -> bool:
        """Round bookkeeping; returns True when the experiment is over."""
        self._wait_neighbors_ready()
        for nc in self.get_neighbors():
            if nc not in self._initial_neighbors:
                self.rm_neighbor(nc)
        with self._round_lock:
            self.aggregator.clear()
            if self.round is not None:
                self.round += 1
                self._aggregation_done.clear()
        self.learner.finalize_round()
        if self.round is None:
            return True
        if self.learner.logger is not None:
            self.learner.logger.log_metrics({"Round": self.round}, step=self.learner.logger.global_step)
        self._train_set = []     # per-peer MODELS_AGGREGATED lists are reset when the peer's MODELS_READY arrives
        if self.checkpoint_dir:
            self.save_checkpoint(os.path.join(self.checkpoint_dir, f"participant_{self.idx}_round_{self.round}.ckpt"))
        # only the node that LED this round hands the role on (a node that received TRANSFER_LEADERSHIP while
        # finishing the same round keeps it for the next one)
        if (self._p("scenario_args")["federation"] == "SDFL" and self.role == Role.AGGREGATOR
                and getattr(self, "_round_role", None) == Role.AGGREGATOR):
            self._transfer_aggregator_role("random")
        logging.info("[NODE] Round %s of %s finished.", self.round, self.totalrounds)
        if self.round < self.totalrounds:
            return False
        self._evaluate()                       # at the end every node computes metrics (node.py:663)
        logging.info("[NODE] FL experiment finished")
        self.round = None
        self.totalrounds = None
        self._model_initialized = False
        self._init_model_ready.clear()
        return True

    def _transfer_aggregator_role(self, schema):
        if schema != "random":
            logging.info("[NODE] leadership schema %s not found", schema)
            return
        nbs = self.get_neighbors()
        if not nbs:
            return
        nc = random.choice(nbs)
        nc.send(CommunicationProtocol.build_transfer_leadership_msg())
        self.config.participant["device_args"]["role"] = Role.TRAINER
        logging.info("[NODE] Aggregator role transferred to %s", nc.get_name())

    # ------------------------------------------------------------------ model gossip (data plane)
    def _gossip_model_aggregation(self):
        # a peer that already announced MODELS_READY for this round needs nothing more
        # The reference compares COUNTS (len(peer's models) < len(my train set), node.py:695). With unequal
        # train sets (star: a trainer's is {server, self}, the server's is everyone) a trainer then believes
        # the server is complete as soon as another trainer was faster, never sends its own model and the
        # round only ends at AGGREGATION_TIMEOUT - so a peer that lacks OUR contribution stays a candidate.
        me = self.get_name()
        cand = lambda nc: (nc.get_name() in self._train_set and nc.get_model_ready_status() < (self.round or 0)
                           and (len(nc.get_models_aggregated()) < len(self._train_set)
                                or me not in nc.get_models_aggregated()))
        status = lambda nc: (nc.get_name(), len(nc.get_models_aggregated()))

        def model(nc):
            # Our own contribution goes out on its own first: the peer certainly wants it (we are its neighbour),
            # whereas a partial aggregate that mixes it with models from outside the peer's train set is refused
            # there (add_model). Afterwards: the reference's partial aggregation of everything the peer lacks.
            have = nc.get_models_aggregated()
            if me not in have:
                own = self.aggregator.get_single(me)
                if own is not None:
                    return own
            return self.aggregator.get_partial_aggregation(have)
        self._gossip_model(cand, status, model)

    def _gossip_model_difusion(self, initialization=False):
        if initialization:
            self._init_model_ready.wait()
            cand = lambda nc: not nc.get_model_initialized()
        else:
            self._aggregation_done.wait(float(self._p("AGGREGATION_TIMEOUT")) + 5.0)
            cand = lambda nc: nc.get_model_ready_status() < self.round
        status = lambda nc: nc.get_name()
        model = lambda _: (self.learner.get_parameters(), None, None)
        self._gossip_model(cand, status, model)

    def _gossip_model(self, candidate_condition, status_function, model_function):
        period = 1.0 / float(self._p("GOSSIP_MODELS_FREC"))
        exit_rounds = int(self._p("GOSSIP_EXIT_ON_X_EQUAL_ROUNDS"))
        per_round = int(self._p("GOSSIP_MODELS_PER_ROUND"))
        block = int(self._p("BLOCK_SIZE"))
        history = []
        while True:
            begin = time.time()
            if self.round is None:
                return
            nei = [nc for nc in self.get_neighbors() if candidate_condition(nc)]
            if not nei:
                return
            history.append(str([status_function(nc) for nc in nei]))
            if len(history) > exit_rounds:
                history.pop(0)
                if all(h == history[0] for h in history):
                    logging.info("[NODE] Gossiping exited for %d equal rounds.", exit_rounds)
                    return
            for nc in random.sample(nei, min(per_round, len(nei))):
                model, contributors, weight = model_function(nc)
                if model is None:
                    continue
                encoded = self.learner.encode_parameters(params=model, contributors=contributors, weight=weight)
                nc.send_many(CommunicationProtocol.build_params_msg(encoded, block))
            sleep = period - (time.time() - begin)
            if sleep > 0:
                time.sleep(sleep)

    # ------------------------------------------------------------------ events
    def update(self, event, obj):
        if event == Events.NODE_CONNECTED_EVENT:
            n, force = obj
            if self.round is not None and not force:
                logging.info("[NODE] Cant connect to other nodes when learning is running.")
                n.stop()
                return
        elif event == Events.SEND_ROLE_EVENT:
            self.broadcast(CommunicationProtocol.build_role_msg(self.get_name(), self.role))
        elif event == Events.ROLE_RECEIVED_EVENT:
            if self.heartbeater is not None:
                self.heartbeater.add_node_role(obj[0], obj[1])
        elif event == Events.AGGREGATION_FINISHED_EVENT:
            if self._aggregation_done.is_set():
                return          # second delivery of a round (own aggregation after an adopted diffusion): keep the first
            if obj is not None:
                self.learner.set_parameters(obj)
                if self.round is not None and not self._ready_sent:
                    self._ready_sent = True
                    self.broadcast(CommunicationProtocol.build_models_ready_msg(self.round))
            else:
                logging.error("[NODE] Aggregation finished with no parameters")
            self._aggregation_done.set()
        elif event == Events.START_LEARNING_EVENT:
            self._start_learning_thread(obj[0], obj[1])
        elif event == Events.STOP_LEARNING_EVENT:
            self._stop_learning()
        elif event == Events.PARAMS_RECEIVED_EVENT:
            self._model_queue.put(obj)          # heavy work happens on the worker, not the socket thread
        elif event == Events.METRICS_RECEIVED_EVENT:
            pass
        elif event == Events.TRAIN_SET_VOTE_RECEIVED_EVENT:
            node, votes = obj
            with self._train_set_votes_lock:
                self._train_set_votes[node] = votes
        elif event == Events.REPORT_STATUS_TO_CONTROLLER_EVENT:
            self._report_status_to_controller()
            self._report_resources()
        elif event == Events.STORE_MODEL_PARAMETERS_EVENT:
            if obj is not None:
                self._stored_model_parameters.append(obj)
                self.broadcast(CommunicationProtocol.build_models_ready_msg(self.round))
            self._aggregation_done.set()
        super().update(event, obj)

    # ------------------------------------------------------------------ controller / resources
    def _report_status_to_controller(self):
        """POST the live participant config to the controller (reference node.py:916-937)."""
        sa = self.config.participant["scenario_args"]
        ctrl = sa.get("controller")
        if not ctrl or not sa.get("name"):
            return
        url = f"http://{ctrl}/scenario/{sa['name']}/node/update"
        try:
            import requests
            r = requests.post(url, data=json.dumps(self.config.participant),
                              headers={"Content-Type": "application/json",
                                       "X-Scenario-Token": str(sa.get("controller_token") or "")}, timeout=2)
            if r.status_code != 200:
                logging.debug("Error received from controller: %s", r.status_code)
        except Exception:
            logging.debug("Error connecting to the controller at %s", url)

    def _report_resources(self):
        """CPU / RAM / disk / network and per-GPU utilisation + memory (reference node.py:939-997 uses
        pynvml; on MI355X the numbers come from sysfs ``gpu_busy_percent`` and ``torch.cuda.mem_get_info``)."""
        logger = self.learner.logger
        if logger is None or not hasattr(logger, "log_metrics_direct"):
            return
        try:
            start = datetime.strptime(self._p("scenario_args")["start_time"], "%d/%m/%Y %H:%M:%S")
            step = int((datetime.now() - start).total_seconds())
        except Exception:
            step = int(time.time())
        try:
            import psutil
            net = psutil.net_io_counters()
            m = {"Resources/CPU_percent": psutil.cpu_percent(), "Resources/RAM_percent": psutil.virtual_memory().percent,
                 "Resources/Disk_percent": psutil.disk_usage("/").percent, "Resources/Bytes_sent": net.bytes_sent,
                 "Resources/Bytes_recv": net.bytes_recv, "Resources/Packets_sent": net.packets_sent,
                 "Resources/Packets_recv": net.packets_recv, "Resources/Uptime": psutil.boot_time()}
            try:
                m["Resources/CPU_temp"] = psutil.sensors_temperatures()["coretemp"][0].current
            except Exception:
                m["Resources/CPU_temp"] = 0
            from .utils.env import gpu_stats
            for i, g in enumerate(gpu_stats()):
                m[f"Resources/GPU{i}_percent"] = g.get("busy_percent", 0)
                m[f"Resources/GPU{i}_temp"] = g.get("temp_c", 0)
                m[f"Resources/GPU{i}_mem_percent"] = g.get("mem_percent", 0)
            logger.log_metrics_direct(m, step)
        except Exception as e:
            logging.debug("resource report failed: %s", e)
