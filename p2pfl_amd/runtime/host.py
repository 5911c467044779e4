"""Control plane of co-resident (virtual) federation nodes: node façades, command bus, timer wheel, TCP gateway.

A :class:`~p2pfl_amd.runtime.federation.BatchedFederation` trains hundreds of nodes as rows of one slab. This
module gives those rows the *node* surface of the reference and keeps heartbeats and the command channel on
the host, as the design calls for:

* :class:`VirtualNode` - a lightweight view (row index + role + neighbour set + protocol state) with the public
  attributes / methods external code uses on a reference ``Node``: ``round``, ``totalrounds``, ``role``,
  ``get_name``, ``get_neighbors``, ``get_network_nodes``, ``set_start_learning``, ``set_stop_learning``,
  ``connect_to``, ``learner``-style ``get_parameters`` (fedstellar/node.py:77-89,103,224-271).
* :class:`CommandBus` - ONE in-process queue of wire-format control messages (the very strings
  ``CommunicationProtocol.build_*_msg`` builds: ``BEAT``, ``ROLE``, ``START_LEARNING``, ``STOP_LEARNING``,
  ``MODELS_AGGREGATED``, ``MODELS_READY``) published per virtual node and delivered to subscribers by a single
  dispatcher - instead of N x (heartbeater + gossiper + per-peer socket threads).
* :class:`TimerWheel` - hashed timer wheel, O(1) per tick: every virtual node's heartbeat (``BEAT`` each
  ``HEARTBEAT_PERIOD``, ``ROLE`` + status POST every second beat, heartbeater.py:66-78) hangs on one wheel driven
  by one service thread.
* :class:`GatewayNode` - a real TCP ``Node`` (this repo's wire-compatible implementation of
  fedstellar/node.py) that *is* one of the virtual nodes towards the outside: a reference-protocol ``Node`` in
  another process / on another machine connects to it, exchanges ``PARAMS`` frames with it and can start / stop
  the whole batched run with ``START_LEARNING`` / ``STOP_LEARNING``. Its learner (:class:`VirtualRowLearner`)
  maps ``PARAMS`` payloads <-> the slab row; the contributions of the node's co-resident neighbours enter its
  aggregator as ONE partial aggregate computed by the reduce kernel (what a neighbour's
  ``get_partial_aggregation`` would send, aggregator.py:181-208).
* :class:`FederationHost` - owns all of the above plus the single learning thread that runs the rounds.

Round hand-shake between the batched loop and a gateway (both see the same round number):

    host:     evaluate + train all rows ......... release "trained(r)" | wait gateway aggregation(r) | FedAvg of
                                                                        |                            | the rows
    gateway:  _train_step(r): wait "trained(r)" -> inject virtual partial aggregate + own row -> TCP gossip with the
              external peers -> aggregation finished (row + external models) ------^
    host:     ... then writes the gateway's aggregate into its row (and into the rows of co-resident nodes that adopt
              it: trainers of a CFL star whose server is the gateway).
"""
from __future__ import annotations

import collections
import json
import logging
import threading
import time
from collections import OrderedDict
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from ..communication_protocol import CommunicationProtocol
from ..config import Config
from ..learning import codec
from ..learning.exceptions import DecodingParamsError, ModelNotMatchingError
from ..learning.learner import NodeLearner
from ..node import Node
from ..role import Role
from ..utils.observer import Events
from .aggregation import build_plan, run_plan
from .federation import BatchedFederation


# ======================================================================================================
# command bus + timer wheel
# ======================================================================================================
class CommandBus:
    """In-process command channel of the virtual nodes. ``publish`` may be called from any thread; ``pump`` (the
    host's service thread) hands every pending message to every subscriber, in order."""

    def __init__(self):
        self._q: "collections.deque" = collections.deque()
        self._subs: List[tuple] = []
        self._lock = threading.Lock()
        self.counts: "collections.Counter" = collections.Counter()     # header -> messages published

    def subscribe(self, fn: Callable[[int, bytes], None], local_only: bool = False) -> None:
        """``local_only``: not for messages that came in from another rank's bus (what forwards to the ranks)."""
        self._subs.append((fn, local_only))

    def publish(self, sender: int, msg: bytes, remote: bool = False) -> None:
        """``remote``: the message was published on another rank's bus and arrived over the rank bridge."""
        with self._lock:
            self._q.append((sender, msg, remote))
            self.counts[msg.split(b" ", 1)[0].strip().decode(errors="replace")] += 1

    def pump(self) -> int:
        n = 0
        while True:
            with self._lock:
                if not self._q:
                    return n
                sender, msg, remote = self._q.popleft()
            for fn, local_only in list(self._subs):
                if remote and local_only:
                    continue
                try:
                    fn(sender, msg)
                except Exception as e:        # a faulty observer must not stop the control plane
                    logging.exception(e)
            n += 1


class TimerWheel:
    """Hashed timer wheel: ``schedule`` and expiry are O(1) per timer and per tick whatever the number of nodes."""

    def __init__(self, tick: float = 0.05, slots: int = 256):
        self.tick, self.nslots = float(tick), int(slots)
        self._slots: List[list] = [[] for _ in range(self.nslots)]
        self._now_tick = 0
        self._lock = threading.Lock()

    def schedule(self, delay: float, fn: Callable[[], None], period: Optional[float] = None) -> None:
        ticks = max(1, int(np.ceil(delay / self.tick - 1e-9)))
        with self._lock:
            due = self._now_tick + ticks
            self._slots[due % self.nslots].append([due, fn, period])

    def advance(self, ticks: int = 1) -> int:
        """Move the wheel forward; fires (and re-arms periodic) timers that came due. Returns #fired."""
        fired = 0
        for _ in range(ticks):
            with self._lock:
                self._now_tick += 1
                slot = self._slots[self._now_tick % self.nslots]
                due_now = [t for t in slot if t[0] <= self._now_tick]
                slot[:] = [t for t in slot if t[0] > self._now_tick]
            for due, fn, period in due_now:
                try:
                    fn()
                except Exception as e:
                    logging.exception(e)
                fired += 1
                if period:
                    self.schedule(period, fn, period)
        return fired


# ======================================================================================================
# node façade
# ======================================================================================================
class VirtualNode:
    """The reference ``Node`` surface of one slab row (fedstellar/node.py:77-89,103,224-271)."""

    def __init__(self, host: "FederationHost", idx: int):
        self._host, self.idx = host, idx

    # identity / state
    def get_name(self) -> str:
        return self._host.names[self.idx]

    @property
    def round(self):
        return self._host.round

    @property
    def totalrounds(self):
        return self._host.totalrounds

    @property
    def role(self) -> str:
        return self._host.fed.roles[self.idx]

    def is_alive(self) -> bool:
        return bool(self._host.fed.alive[self.idx])

    # topology / membership
    def get_neighbors(self) -> List[str]:
        h = self._host
        names = [h.names[j] for j in h.fed.neighbors(self.idx)]
        gw = h.gateways.get(self.idx)
        if gw is not None:
            names += gw.get_neighbors_names()
        return names

    def get_network_nodes(self) -> List[str]:
        return self._host.network_nodes()

    def is_local(self) -> bool:
        """Does THIS rank host the node's row (model access and gateways need that; everything else works anywhere)?"""
        return self.idx in self._host.fed.local

    def connect_to(self, h, p, full=False, force=False):
        """Peer with an external TCP node (through this node's gateway endpoint, created on demand)."""
        return self._host.gateway(self.idx).connect_to(h, p, full, force)

    # learning control
    def set_start_learning(self, rounds: int = 1, epochs: int = 1) -> None:
        self._host.start_learning(rounds, epochs, initiator=self.idx)

    def set_stop_learning(self) -> None:
        self._host.stop_learning()

    # model access (learner-style)
    def get_parameters(self) -> "OrderedDict[str, torch.Tensor]":
        if not self.is_local():
            raise LookupError(f"node {self.idx} lives on rank {int(self._host.fed.owner[self.idx])}, not on rank "
                              f"{self._host.fed.rank}: ask that rank's host for its parameters")
        return self._host.fed.engine.get_state_dict(self.idx - self._host.fed.local.start)

    def get_num_samples(self):
        return int(self._host.fed.weights[self.idx]), self._host.fed.engine.num_samples()[1]


# ======================================================================================================
# gateway: a virtual node seen from the TCP side
# ======================================================================================================
class VirtualRowLearner(NodeLearner):
    """Learner of a :class:`GatewayNode`: the model is a slab row, training happens in the batched engine."""

    def __init__(self, model=None, data=None, config=None, logger=None):
        self.logger = None
        self.host: Optional["FederationHost"] = None
        self.idx = -1
        self._sd: Optional[OrderedDict] = None        # host copy of what this node currently holds
        self._cv = threading.Condition()
        self._trained_round = -1
        self._interrupted = False

    def bind(self, host: "FederationHost", idx: int) -> None:
        self.host, self.idx = host, idx

    # -- payload codec (the reference format, lightninglearner.py:73-89)
    @property
    def _layout(self):
        return self.host.fed.engine.layout

    def encode_parameters(self, params=None, contributors=None, weight=None):
        sd = params if params is not None else self.get_parameters()
        return codec.encode_parameters([sd[s.name].detach().cpu().numpy() for s in self._layout.slots],
                                       contributors, weight)

    def decode_parameters(self, data):
        arrays, contributors, weight = codec.decode_parameters(data)
        slots = self._layout.slots
        if len(arrays) != len(slots):
            raise DecodingParamsError(f"expected {len(slots)} tensors, got {len(arrays)}")
        sd = OrderedDict((s.name, torch.as_tensor(np.asarray(a), dtype=torch.float32)) for s, a in zip(slots, arrays))
        return sd, contributors, weight

    def check_parameters(self, params):
        return self._layout.check_state_dict(params)

    def set_parameters(self, params):
        if not self.check_parameters(params):
            raise ModelNotMatchingError("Not matching models")
        self._sd = OrderedDict((k, v.detach().to("cpu", torch.float32).clone()) for k, v in params.items())
        self.host.on_gateway_parameters(self.idx, self._sd)

    def get_parameters(self):
        if self._sd is None:
            self._sd = self._row_state_dict()
        return self._sd

    def _row_state_dict(self):
        eng = self.host.fed.engine
        sd = eng.get_state_dict(self.idx - self.host.fed.local.start)
        return OrderedDict((k, v.detach().cpu()) for k, v in sd.items())

    # -- training happens elsewhere: wait for it
    def release_trained(self, rnd: int) -> None:
        with self._cv:
            self._trained_round = max(self._trained_round, rnd)
            self._cv.notify_all()

    def wait_trained(self, rnd: int, timeout: float) -> bool:
        with self._cv:
            ok = self._cv.wait_for(lambda: self._trained_round >= rnd or self._interrupted, timeout)
        if ok and not self._interrupted:
            self._sd = self._row_state_dict()         # this node's own (trained) model of the round
        return ok and not self._interrupted

    def fit(self):
        node = self.host.gateways.get(self.idx)
        rnd = node.round if node is not None and node.round is not None else 0
        self.wait_trained(rnd, self.host.aggregation_timeout * 4)

    def interrupt_fit(self):
        with self._cv:
            self._interrupted = True
            self._cv.notify_all()

    def evaluate(self):
        return None                                   # the host evaluates all rows at once

    def get_num_samples(self):
        return int(self.host.fed.weights[self.idx]), self.host.fed.engine.num_samples()[1]

    def init(self):
        with self._cv:
            self._interrupted = False

    def set_epochs(self, epochs):
        pass

    def create_trainer(self):
        pass

    def finalize_round(self):
        pass

    def close(self):
        pass

    def set_model(self, model):
        pass

    def set_data(self, data):
        pass


class GatewayNode(Node):
    """TCP endpoint of virtual node ``vidx``: towards external peers it behaves exactly like a ``Node`` with that
    node's name and role; towards the host it is the place where external models meet the slab."""

    def __init__(self, host: "FederationHost", vidx: int, listen_host: str, port: int, config: Config):
        self._vhost, self.vidx = host, vidx
        Node.__init__(self, vidx, host.scenario, None, None, host=listen_host, port=port, config=config,
                      learner=VirtualRowLearner)
        self.learner.bind(host, vidx)

    @property
    def role(self):                 # the role lives in the federation (SDFL moves it)
        return self._vhost.fed.roles[self.vidx]

    def _validate_train_set(self):
        # co-resident neighbours are neighbours too: their models arrive as one partial aggregate (_snapshot_model)
        for name in self._vhost.virtual_train_set_names(self.vidx):
            if name not in self._train_set:
                self._train_set.append(name)
        live = set(self.get_network_nodes()) | set(self._vhost.network_nodes())
        self._train_set = [n for n in self._train_set if n in live or n == self.get_name()]
        logging.info("[GATEWAY] Train set of %d nodes: %s", len(self._train_set), self._train_set)

    def _snapshot_model(self):
        """Called by the round loop right before this node adds its own model: the batched training of the round
        must be over (servers never call ``fit``), then the co-resident contributions go in first."""
        rnd = self.round if self.round is not None else 0
        self.learner.wait_trained(rnd, self._vhost.aggregation_timeout * 4)
        part = self._vhost.virtual_partial(self.vidx)
        if part is not None and self.round is not None:
            sd, names, weight = part
            added = self.aggregator.add_model(sd, names, weight)
            if added is not None:
                self.broadcast(CommunicationProtocol.build_models_aggregated_msg(added))
        return Node._snapshot_model(self)

    def update(self, event, obj):
        if event == Events.START_LEARNING_EVENT:
            self._vhost.on_external_start(int(obj[0]), int(obj[1]), self.vidx)
        elif event == Events.STOP_LEARNING_EVENT:
            self._vhost.on_external_stop(self.vidx)
        elif event == Events.BEAT_RECEIVED_EVENT:
            self._vhost.on_external_beat(obj)
        Node.update(self, event, obj)


# ======================================================================================================
# the host
# ======================================================================================================
class FederationHost:
    """Control plane of one :class:`BatchedFederation` - one host per RANK for the nodes that rank hosts.

    Several ranks (one process per GPU): every host beats, reports and serves gateways for ITS nodes; the hosts are
    joined by a :class:`~p2pfl_amd.runtime.bridge.RankBridge` (TCP between the rank hosts, never the RCCL stream) that
    carries each bus message to the other ranks' buses - so a gateway on any rank floods the BEAT / ROLE of all N nodes
    to its TCP peers (heartbeater.py:66-139) - and START / STOP_LEARNING to every rank's learning loop
    (node.py:224-271). The loops run the collective exchange in lock step, so what must be identical on all ranks is
    AGREED at the top of every round over the job's host-side gloo group: the stop flag (all ranks leave the loop at
    the same round) and the set of gateway nodes (a gateway's rank needs the ROWS of its train set, so the exchange
    must not collapse into an all-reduce; its aggregate goes back to adopting nodes on other ranks by a broadcast)."""

    def __init__(self, fed: BatchedFederation, names: Optional[List[str]] = None, scenario: str = "scenario",
                 participant: Optional[dict] = None, tick: float = 0.05, status_poster: Optional[Callable] = None):
        self.fed = fed
        self.scenario = scenario
        base = Config(participant=participant or {}).participant
        self.participant = base
        self.heartbeat_period = float(base["HEARTBEAT_PERIOD"])
        self.node_timeout = float(base["NODE_TIMEOUT"])
        self.aggregation_timeout = float(base["AGGREGATION_TIMEOUT"])
        self.names = list(names) if names is not None else [f"127.0.0.1:{45000 + i}" for i in range(fed.N)]
        self.nodes = [VirtualNode(self, i) for i in range(fed.N)]
        self.bus = CommandBus()
        self.wheel = TimerWheel(tick)
        self.gateways: Dict[int, GatewayNode] = {}
        self.external: Dict[str, float] = {}           # external node name -> last beat (wall clock)
        self.round: Optional[int] = None
        self.totalrounds: Optional[int] = None
        self.epochs = fed.epochs
        self.reports = []
        self.status_poster = status_poster             # fn(node idx, participant dict) for the controller POST
        self.round_callback: Optional[Callable] = None # fn(RoundReport) after every round (learning thread)
        self._stop = threading.Event()                 # stop the learning loop
        self._terminate = threading.Event()            # stop the service thread
        self._init_ready = threading.Event()
        self._init_ready.set()                         # rows start identical (fed.init_params) unless an external node leads
        self._lock = threading.RLock()
        self._learning_thread: Optional[threading.Thread] = None
        self._service: Optional[threading.Thread] = None
        self._beats = np.zeros(fed.N, dtype=np.int64)
        self._last_tick = None
        self.rank, self.world = fed.rank, fed.world
        self._init_src: Optional[int] = None           # rank whose gateway received an external initiator's model
        self._gw_mask = np.zeros(fed.N, dtype=np.uint8)  # gateway nodes of the WHOLE job (agreed per round)
        self.bridge = None
        if fed.world > 1:
            from .bridge import RankBridge
            group = fed.ensure_host_group()             # collective: every rank builds its host at the same point
            if group is None:
                raise ValueError("a multi-rank FederationHost needs the federation's process group")
            self.bridge = RankBridge(fed.rank, fed.world, group, self._on_bridge_frame)
            self.bus.subscribe(self._forward_to_ranks, local_only=True)
        fed.external_clock = True
        fed.membership.period = self.heartbeat_period
        fed.membership.timeout = self.node_timeout
        fed.aggregation_timeout = self.aggregation_timeout
        fed.pre_aggregate_hook = self._pre_aggregate
        fed.post_aggregate_hook = self._post_aggregate
        self.bus.subscribe(self._forward_to_gateways)
        for i in fed.local:                             # this rank's nodes; staggered over one period
            self.wheel.schedule(self.heartbeat_period * (i + 1) / fed.N, lambda i=i: self._beat(i), self.heartbeat_period)

    # ------------------------------------------------------------------ service thread
    def start(self) -> "FederationHost":
        self._last_tick = time.time()
        self._service = threading.Thread(target=self._serve, name="federation-host", daemon=True)
        self._service.start()
        return self

    def wait_started(self, timeout: Optional[float] = None) -> bool:
        """Block until learning runs (a START_LEARNING from a gateway's peer, or :meth:`start_learning`)."""
        end = None if timeout is None else time.time() + timeout
        while self._learning_thread is None:
            if end is not None and time.time() > end:
                return False
            time.sleep(0.05)
        return True

    def wait_finished(self, timeout: Optional[float] = None) -> bool:
        """Block until the learning thread has ended (all rounds done or STOP_LEARNING)."""
        t = self._learning_thread
        if t is None:
            return True
        t.join(timeout)
        return not t.is_alive()

    def stop(self) -> None:
        self.stop_learning()
        self._terminate.set()
        if self._service is not None and self._service is not threading.current_thread():
            self._service.join(timeout=5)
        self.bus.pump()          # what was published after the service thread's last turn still reaches gateways / ranks
        for gw in list(self.gateways.values()):
            gw.stop()
        if self._learning_thread is not None:
            self._learning_thread.join(timeout=10)
        if self.bridge is not None:
            self.bridge.close()

    def _serve(self) -> None:
        while not self._terminate.is_set():
            self.tick_once()
            self._terminate.wait(self.wheel.tick)

    def tick_once(self, now: Optional[float] = None) -> None:
        """One turn of the control plane (the service thread's body; callable directly with a fake clock)."""
        now = time.time() if now is None else now
        last = self._last_tick if self._last_tick is not None else now
        ticks = max(1, int((now - last) / self.wheel.tick))
        self._last_tick = last + ticks * self.wheel.tick
        self.wheel.advance(ticks)
        self.fed.membership.advance(ticks * self.wheel.tick)
        for name, t in list(self.external.items()):
            if now - t > self.node_timeout:
                self.external.pop(name, None)
        self.bus.pump()

    # ------------------------------------------------------------------ heartbeats (timer wheel -> bus)
    def _beat(self, i: int) -> None:
        m = self.fed.membership
        if not m.beating[i]:
            return                                      # a killed / paused node is silent: NODE_TIMEOUT will evict it
        m.last_beat[i] = m.clock
        self.bus.publish(i, CommunicationProtocol.build_beat_msg(self.names[i]))
        self._beats[i] += 1
        if self._beats[i] % 2 == 0:                     # every second beat: ROLE + status report (heartbeater.py:75-78)
            self.bus.publish(i, CommunicationProtocol.build_role_msg(self.names[i], self.fed.roles[i]))
            if self.status_poster is not None:
                self.status_poster(i, self.participant_view(i))

    def participant_view(self, i: int) -> dict:
        """Live participant config of virtual node i (what the reference POSTs to the controller, node.py:916-937)."""
        p = json.loads(json.dumps(self.participant))
        ip, port = self.names[i].rsplit(":", 1)
        p["device_args"].update(idx=i, role=self.fed.roles[i])
        p["network_args"].update(ip=ip, port=int(port), neighbors=" ".join(self.nodes[i].get_neighbors()))
        p["scenario_args"].update(name=self.scenario, n_nodes=self.fed.N, federation=self.fed.federation)
        return p

    def _forward_to_gateways(self, sender: int, msg: bytes) -> None:
        """BEAT / ROLE of co-resident nodes reach external peers through the gateways (flooded on from there), so an
        external node's heartbeat table lists the virtual nodes like any other member of the network."""
        head = msg.split(b" ", 1)[0]
        if head not in (b"BEAT", b"ROLE"):
            return
        for vidx, gw in self.gateways.items():
            if sender != vidx:                          # the gateway's own Node sends its own beats
                gw.broadcast(msg)

    # ------------------------------------------------------------------ other ranks (host-side bridge)
    def _forward_to_ranks(self, sender: int, msg: bytes) -> None:
        """Every message published on THIS rank's bus also appears on the other ranks' buses."""
        from .bridge import BUS
        self.bridge.send_all(BUS, sender, msg)

    def _on_bridge_frame(self, src_rank: int, kind: int, sender: int, payload: bytes) -> None:
        """A frame of another rank's host (bridge reader thread)."""
        from . import bridge as B
        if kind == B.BUS:
            if payload.startswith(b"BEAT") and 0 <= sender < self.fed.N:
                m = self.fed.membership
                m.last_beat[sender] = max(m.last_beat[sender], m.clock)
            self.bus.publish(sender, payload, remote=True)
        elif kind == B.START:
            d = json.loads(payload.decode())
            self._remote_start(int(d["rounds"]), int(d["epochs"]), int(d["init_src"]))
        elif kind == B.STOP:
            self._local_stop(via=-1)
        elif kind == B.NAME:
            if 0 <= sender < self.fed.N:
                self.names[sender] = payload.decode()
        elif kind == B.XBEAT:
            name = payload.decode()
            if name not in self.names:
                self.external[name] = time.time()

    # ------------------------------------------------------------------ membership views
    def network_nodes(self) -> List[str]:
        live = [self.names[i] for i in np.nonzero(self.fed.membership.believed())[0]]
        return live + [n for n in self.external if n not in live]

    def on_external_beat(self, name: str) -> None:
        if name not in self.names:
            self.external[name] = time.time()
            if self.bridge is not None:
                from .bridge import XBEAT
                self.bridge.send_all(XBEAT, -1, name.encode())

    def virtual_train_set_names(self, i: int) -> List[str]:
        return [self.names[j] for j in self.fed.train_set(i) if j != i]

    # ------------------------------------------------------------------ gateways
    def gateway(self, vidx: int, listen_host: str = "127.0.0.1", port: Optional[int] = None) -> GatewayNode:
        """The TCP endpoint of virtual node ``vidx`` (created and started on first use). It listens on the port of
        the node's name unless ``port`` says otherwise (0 = any free port: the node is renamed to it)."""
        with self._lock:
            gw = self.gateways.get(vidx)
            if gw is not None:
                return gw
            if vidx not in self.fed.local:
                raise ValueError(f"virtual node {vidx} is hosted by rank {int(self.fed.owner[vidx])}: its TCP gateway "
                                 f"is opened by that rank's host, not by rank {self.rank}")
            if port is None:
                port = int(self.names[vidx].rsplit(":", 1)[1])
            p = json.loads(json.dumps(self.participant))
            p["device_args"].update(idx=vidx, role=self.fed.roles[vidx])
            p["scenario_args"].update(name="", controller="", federation=self.fed.federation, n_nodes=self.fed.N)
            p["tracking_args"].update(local_tracking="none", log_dir="")
            gw = GatewayNode(self, vidx, listen_host, port, Config(participant=p))
            self.names[vidx] = gw.get_name()
            gw.start()
            self.gateways[vidx] = gw
            self._gw_mask[vidx] = 1
            if self.bridge is not None:
                from .bridge import NAME
                self.bridge.send_all(NAME, vidx, self.names[vidx].encode())
            return gw

    def on_gateway_parameters(self, vidx: int, sd) -> None:
        """The gateway adopted a model. Before round 0 that is the INITIAL model of an external start node: every
        co-resident node starts from it (init-model diffusion, node.py:298-299,372-379)."""
        if not self._init_ready.is_set():
            eng = self.fed.engine
            row = torch.zeros(eng.layout.row_stride)
            eng.layout.state_dict_to_row(sd, row)
            eng.params.copy_(row.to(eng.params.device).unsqueeze(0).expand(eng.K, -1))
            self._init_ready.set()

    def virtual_partial(self, g: int):
        """(state_dict, contributor names, weight) of the virtual part of g's train set without g itself - the partial
        aggregate a neighbour would gossip (aggregator.py:181-208), computed by the reduce kernel over this rank's rows
        and the rows of other ranks that the round's exchange has staged (``fed.aggregate`` runs the exchange before it
        releases the gateways)."""
        fed = self.fed
        members = [j for j in fed.train_set(g) if j != g and fed._present(j)
                   and fed.roles[j] not in (Role.PROXY, Role.IDLE) and (Role.trains(fed.roles[j]) or Role.aggregates(fed.roles[j]))]
        lo = fed.local.start
        row_of = fed._row_of or {j: j - lo for j in fed.local}
        missing = [j for j in members if j not in row_of]
        if missing:
            # (a gateway opened in the middle of a round, or one whose node does not aggregate: the exchange of this
            # round did not stage those rows)
            logging.warning("[HOST] gateway %d: rows of %s are not on rank %d this round; partial aggregate without them",
                            g, missing, self.rank)
            members = [j for j in members if j in row_of]
        if not members:
            return None
        eng = fed.engine
        out = torch.zeros(1, eng.layout.row_stride, device=eng.params.device)
        plan = build_plan([([row_of[j] for j in members], [float(fed.weights[j]) for j in members], [0])])
        run_plan(eng.params, fed.staging, out, plan)
        sd = OrderedDict((k, v.detach().cpu()) for k, v in eng.layout.row_to_state_dict(out[0]).items())
        return sd, [self.names[j] for j in members], int(sum(fed.weights[j] for j in members))

    # ------------------------------------------------------------------ learning control
    def start_learning(self, rounds: int, epochs: int, initiator: Optional[int] = None) -> None:
        """``Node.set_start_learning`` of a virtual node (node.py:224-257): START_LEARNING goes out on the bus and,
        through the gateways, to external peers, which also receive the initial model from them. The learning loops
        of ALL ranks start (bridge)."""
        with self._lock:
            if self.round is not None:
                logging.info("[HOST] Learning already started")
                return
            self._init_src = None
            self._begin(rounds, epochs)
        self._announce_start(rounds, epochs, initiator or 0, -1)
        for gw in self.gateways.values():
            if gw.round is None:
                gw.set_start_learning(rounds, epochs)

    def on_external_start(self, rounds: int, epochs: int, via: int) -> None:
        """START_LEARNING arrived over TCP: the external node is the initiator, its model is the initial one."""
        with self._lock:
            if self.round is not None:
                return
            self._init_ready.clear()                    # wait for the initiator's weights (on_gateway_parameters)
            self._init_src = self.rank if self.world > 1 else None
            self._begin(rounds, epochs)
        self._announce_start(rounds, epochs, via, self.rank)
        for vidx, gw in self.gateways.items():
            if vidx != via and gw.round is None:
                gw._start_learning_thread(rounds, epochs)

    def _announce_start(self, rounds: int, epochs: int, sender: int, init_src: int) -> None:
        self.bus.publish(sender, CommunicationProtocol.build_start_learning_msg(rounds, epochs))
        if self.bridge is not None:
            from .bridge import START
            self.bridge.send_all(START, sender, json.dumps({"rounds": rounds, "epochs": epochs,
                                                            "init_src": init_src}).encode())

    def _remote_start(self, rounds: int, epochs: int, init_src: int) -> None:
        """Another rank's host started learning (a virtual node there, or an external initiator at one of its
        gateways whose model that rank will broadcast: ``init_src``)."""
        with self._lock:
            if self.round is not None:
                return
            self._init_src = init_src if init_src >= 0 else None
            self._begin(rounds, epochs)
        for gw in self.gateways.values():
            if gw.round is None:
                if init_src < 0:
                    gw.set_start_learning(rounds, epochs)       # its TCP peers get START_LEARNING + the initial model
                else:
                    gw._start_learning_thread(rounds, epochs)

    def _begin(self, rounds: int, epochs: int) -> None:
        self.round, self.totalrounds, self.epochs = 0, int(rounds), int(epochs)
        self.fed.epochs = int(epochs)
        self._stop.clear()
        t = threading.Thread(target=self._learn, name="federation-learning", daemon=True)
        t.start()
        self._learning_thread = t          # published only once started: wait_started() / wait_finished() may join it

    def stop_learning(self) -> None:
        """``Node.set_stop_learning`` of a virtual node: the run ends after the round in flight - on every rank (the
        loops agree on the stop flag at the top of each round)."""
        if self.round is None:
            return
        self._stop.set()
        self.bus.publish(0, CommunicationProtocol.build_stop_learning_msg())
        if self.bridge is not None:
            from .bridge import STOP
            self.bridge.send_all(STOP, -1)
        for gw in self.gateways.values():
            if gw.round is not None:
                gw.set_stop_learning()

    def on_external_stop(self, via: int) -> None:
        if self.round is None:
            return
        if self.bridge is not None:
            from .bridge import STOP
            self.bridge.send_all(STOP, via)
        self._local_stop(via)

    def _local_stop(self, via: int) -> None:
        """STOP_LEARNING that did not originate at one of this host's virtual nodes (a TCP peer of gateway ``via``,
        another rank, or the ranks' agreement at the top of a round)."""
        if self.round is None or self._stop.is_set():
            return
        self._stop.set()
        self._init_ready.set()
        self.bus.publish(max(via, 0), CommunicationProtocol.build_stop_learning_msg(), remote=via < 0)
        for vidx, gw in self.gateways.items():
            gw.learner.interrupt_fit()
            if vidx != via and gw.round is not None:
                gw.set_stop_learning()

    def _agree(self) -> bool:
        """Top of a round, all ranks in lock step: OR of the stop flags and of the gateway masks over the job's
        host-side group. Returns the agreed stop flag."""
        fed = self.fed
        if self.world == 1:
            return self._stop.is_set()
        import torch.distributed as dist
        t = torch.zeros(1 + fed.N, dtype=torch.uint8)
        t[0] = 1 if self._stop.is_set() else 0
        t[1:] = torch.from_numpy(self._gw_mask)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=fed.host_pg)
        self._gw_mask = t[1:].numpy().copy()
        fed.allow_allreduce = not bool(self._gw_mask.any())
        stop = bool(t[0])
        if stop and not self._stop.is_set():
            self._local_stop(via=-1)
        return stop

    # ------------------------------------------------------------------ the learning thread
    def _learn(self) -> None:
        fed = self.fed
        try:
            if not self._init_ready.wait(self.aggregation_timeout * 4):
                logging.error("[HOST] the initial model never arrived")
            if self.world > 1 and self._init_src is not None:
                # an external initiator's model reached ONE rank's gateway: its row goes to every rank (the same
                # init-model diffusion, node.py:298-299,702-724, as one broadcast)
                eng = fed.engine
                row = eng.params[0].clone()
                fed.comm.broadcast(row, self._init_src)
                if self.rank != self._init_src:
                    eng.params.copy_(row.unsqueeze(0).expand(eng.K, -1))
            first = fed.round
            while self.round is not None and self.round < self.totalrounds:
                if self._agree():
                    break
                rep = fed.run_round()
                self.reports.append(rep)
                if self.round_callback is not None:
                    self.round_callback(rep)
                for i in fed.local:
                    if fed.alive[i]:
                        self.bus.publish(i, CommunicationProtocol.build_models_ready_msg(self.round))
                self.round = fed.round - first
            if not self._stop.is_set():
                self.final_test = fed.engine.evaluate()     # at the end every node computes metrics (node.py:663)
        except Exception as e:
            logging.exception(e)
        finally:
            for gw in self.gateways.values():
                gw.learner.interrupt_fit()
            self.round = None
            self.totalrounds = None

    # hooks of BatchedFederation.run_round (learning thread)
    def _pre_aggregate(self, fed: BatchedFederation) -> None:
        for i in fed.local:
            if fed._present(i) and fed.roles[i] not in (Role.PROXY, Role.IDLE):
                self.bus.publish(i, CommunicationProtocol.build_models_aggregated_msg([self.names[i]]))
        rnd = self.round if self.round is not None else 0
        for gw in self.gateways.values():
            gw.learner.release_trained(rnd)
        deadline = time.time() + self.aggregation_timeout * 2
        grace = time.time() + float(self.participant["WAIT_HEARTBEATS_CONVERGENCE"]) + 5.0
        for vidx, gw in self.gateways.items():
            if not fed._present(vidx):
                continue
            while not self._stop.is_set() and time.time() < deadline:
                r = gw.round
                if r is None:
                    # its learning thread has not come up yet (START_LEARNING still in flight) - or it is over
                    if rnd > 0 or time.time() > grace:
                        break
                elif r > rnd:
                    break
                elif r == rnd and gw._aggregation_done.wait(0.02):
                    # the gateway's Node ran the same round: ITS aggregate (co-resident + external models) is in
                    break
                time.sleep(0.005)

    def _post_aggregate(self, fed: BatchedFederation) -> None:
        gws = sorted(int(i) for i in np.nonzero(self._gw_mask)[0]) if self.world > 1 else sorted(self.gateways)
        if not gws:
            return
        eng = fed.engine
        lo = fed.local.start
        _, adopt = fed._round_structure()
        for vidx in gws:
            if not fed._present(vidx):
                continue
            owner = int(fed.owner[vidx])
            takers = [w for w, a in adopt.items() if a == vidx]
            if owner == self.rank:
                gw = self.gateways.get(vidx)
                if gw is not None and gw.learner._sd is not None:
                    row = torch.zeros(eng.layout.row_stride)
                    eng.layout.state_dict_to_row(gw.learner._sd, row)
                    row = row.to(eng.params.device)
                    eng.params[vidx - lo].copy_(row)
                    for w in takers:                    # co-resident waiting nodes that take the gateway's result
                        if w in fed.local:
                            eng.params[w - lo].copy_(row)
            if self.world > 1 and any(int(fed.owner[w]) != owner for w in takers):
                # waiting nodes on OTHER ranks adopt the gateway's aggregate too (it holds the external models their own
                # rank never saw): one broadcast of the row from the gateway's rank; every rank derives the same
                # ``adopt`` map, so all of them take part
                row = eng.params[vidx - lo].clone() if owner == self.rank else torch.empty_like(eng.params[0])
                fed.comm.broadcast(row, owner)
                if owner != self.rank:
                    for w in takers:
                        if w in fed.local:
                            eng.params[w - lo].copy_(row)
