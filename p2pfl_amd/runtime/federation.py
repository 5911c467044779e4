"""BatchedFederation: hundreds of virtual federation nodes per GPU, one process per GPU.

This is the MI355X-native execution model of the framework. Every rank (= one GPU) hosts a
contiguous block of the scenario's nodes inside one :class:`ShardEngine`; a federated round is

    evaluate  ->  local training (epochs)  ->  model exchange  ->  FedAvg  ->  bookkeeping

which reproduces the per-node round of the reference (``Node.__train_step`` /
``__on_round_finished``, fedstellar/node.py:402-535,632-674) for all nodes at once:

* ``train_set(i)`` = live direct neighbours of i plus i itself (node.py:411-422,537-548);
* aggregators / servers average their train set weighted by training-sample counts
  (node.py:444-448, fedavg.py:40-60); a SERVER never trains (node.py:438);
* trainers / proxies / idle nodes adopt the model of an aggregating neighbour
  (aggregator.py:119-123); in SDFL the aggregator role moves to a random neighbour after every
  round (node.py:648-650,676-686);
* dead nodes (fault injection / heartbeat eviction) drop out of every train set and stop training.

Model exchange is *not* message passing: rows of co-resident neighbours are read in place by
the fused reduce kernel; rows owned by other GPUs are moved with RCCL over xGMI — a single
all-reduce of the local partial sum when every node shares one train set (fully connected, star),
or one variable-split all-to-all of exactly the rows each peer needs (ring / random graphs).
With ``encrypt=True`` the rows that leave a GPU are AES-128-ECB encrypted on the device with a
per-link key and decrypted on arrival (the reference's per-link AES, encrypter.py:109-193).
"""
from __future__ import annotations

import hashlib
import os
import random
import time
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..models.specs import ModelSpec
from ..parallel.collectives import DeviceExchange
from ..parallel.planner import link_key, partition_nodes, plan_exchange  # noqa: F401  (re-exported)
from ..role import Role, default_roles
from .aggregation import AggregationPlan, build_plan, run_plan
from .engine import ShardEngine, make_engine
from .membership import VirtualMembership


@dataclass
class RoundReport:
    round: int
    seconds: float
    phase_seconds: Dict[str, float]
    train_samples: int               # samples processed by local training on this rank
    test_loss: Optional[torch.Tensor] = None
    test_accuracy: Optional[torch.Tensor] = None
    train_loss: Optional[torch.Tensor] = None
    exchanged_rows: int = 0


class BatchedFederation:
    def __init__(self, spec: ModelSpec, topology, *, federation: str = "DFL", roles: Optional[Sequence[str]] = None,
                 device="cpu", engine: str = "auto", batch_size: int = 32, epochs: int = 1, seed: int = 0,
                 rank: int = 0, world_size: int = 1, process_group=None, encrypt: bool = False,
                 scenario: str = "scenario", persist_optimizer: bool = False, use_graph: bool = True,
                 overlap: bool = True, heartbeat_period: float = 4.0, node_timeout: float = 20.0,
                 instant_detection: bool = True, precision: Optional[str] = None):
        self.spec = spec
        self.adj = np.asarray(topology, dtype=np.float32)
        self.N = self.adj.shape[0]
        self.federation = federation.upper()
        self.roles = list(roles) if roles is not None else default_roles(self.federation, self.N)
        self.epochs = epochs
        self.rank, self.world = rank, world_size
        self.pg = process_group
        self.encrypt = encrypt
        self.scenario = scenario
        self.device = torch.device(device)
        self.blocks = partition_nodes(self.N, self.world)
        self.owner = np.zeros(self.N, dtype=np.int64)
        for r, blk in enumerate(self.blocks):
            self.owner[list(blk)] = r
        self.local = self.blocks[rank]
        self.n_local = len(self.local)
        self.engine: ShardEngine = make_engine(spec, self.n_local, self.device, impl=engine, batch_size=batch_size,
                                               seed=seed, persist_optimizer=persist_optimizer,
                                               client_offset=self.local.start,
                                               **({"use_graph": use_graph, "overlap": overlap, "precision": precision}
                                                  if engine != "eager" else {}))
        # membership of ALL nodes of the scenario (every rank applies the same deterministic schedule)
        self.membership = VirtualMembership(self.N, heartbeat_period, node_timeout, instant=instant_detection)
        self.alive = np.ones(self.N, dtype=bool)
        self.weights = np.zeros(self.N, dtype=np.float64)       # training samples per node
        self.round = 0
        self.rng = random.Random(seed)                            # shared by all ranks (same seed)
        self.comm = DeviceExchange(rank, world_size, process_group, self.device, encrypt, scenario)
        self.staging: Optional[torch.Tensor] = None
        self._partial: Optional[torch.Tensor] = None
        self._plan_cache: Dict[bytes, tuple] = {}
        self.trace = os.environ.get("P2PFL_TRACE", "0") not in ("", "0")   # roctx ranges around the phases
        self.reports: List[RoundReport] = []

    # ------------------------------------------------------------------ setup
    def init_params(self, seed: Optional[int] = None, start_node: int = 0) -> None:
        """Initial-model diffusion (node.py:298-299,372-379,702-724): every node of the scenario starts from the
        START node's weights. The rank that owns the start node initialises it; with a process group its row is
        broadcast (RCCL) to all ranks instead of relying on every rank drawing the same numbers."""
        self.engine.init_params(seed=seed, identical=True)
        if self.pg is not None:
            src = int(self.owner[start_node])
            row = self.engine.params[start_node - self.local.start if src == self.rank else 0].clone()
            self.comm.broadcast(row, src)
            self.engine.params.copy_(row.unsqueeze(0).expand(self.n_local, -1))

    def set_local_data(self, train_x, train_y, val_x=None, val_y=None, test_x=None, test_y=None,
                       n_train_all: Optional[Sequence[int]] = None) -> None:
        """Install the data shards of this rank's nodes. ``n_train_all``: training-set size of EVERY
        node of the scenario (FedAvg weights, node.py:447); default: all equal to the local size."""
        self.engine.set_data(train_x, train_y, val_x, val_y, test_x, test_y)
        self._plan_cache.clear()         # plans carry the FedAvg weights
        if n_train_all is None:
            self.weights[:] = train_x.shape[1]
        else:
            self.weights[:] = np.asarray(n_train_all, dtype=np.float64)

    def fast_forward(self, first_round: int) -> None:
        """Resume at ``first_round``: the faults of the rounds already done are applied (with the clock moving as
        if those rounds had run), the round and per-round shuffle counters continue - no training."""
        for r in range(self.round, first_round):
            self.membership.apply_round(r)
            self.membership.advance(self.membership.period)
        self.alive = self.membership.alive().copy()
        self.round = first_round
        self.engine._perm_calls = first_round

    def schedule_fault(self, round_idx: int, action: str, node: int, peer: Optional[int] = None) -> None:
        """Deterministic churn applied at the start of a round: ``kill`` / ``revive`` / ``pause`` / ``resume``
        of a node, ``drop_link`` / ``restore_link`` of the edge (node, peer); see runtime/membership.py."""
        self.membership.schedule(round_idx, action, node, peer)

    # ------------------------------------------------------------------ per-round structure
    def neighbors(self, i: int) -> List[int]:
        return [int(j) for j in np.nonzero(self.adj[i])[0] if j != i and self.membership.link_up(i, int(j))]

    def train_set(self, i: int) -> List[int]:
        if not self.alive[i]:
            return []
        return sorted([j for j in self.neighbors(i) if self.alive[j]] + [i])

    def _round_structure(self):
        """(train_sets of aggregating nodes, adopt map of waiting nodes) for the current state."""
        train_sets: Dict[int, List[int]] = {}
        adopt: Dict[int, int] = {}
        for i in range(self.N):
            if self.alive[i] and Role.aggregates(self.roles[i]):
                # contributors: live members of the train set that hold a model to contribute
                # (proxy/idle nodes never add their own model, node.py:492-524)
                ts = [j for j in self.train_set(i) if self.roles[j] not in (Role.PROXY, Role.IDLE)]
                train_sets[i] = ts
        for i in range(self.N):
            if self.alive[i] and Role.waits(self.roles[i]):
                aggs = [j for j in self.neighbors(i) if self.alive[j] and j in train_sets]
                if aggs:
                    adopt[i] = aggs[0]
        return train_sets, adopt

    def _structure_key(self) -> bytes:
        h = hashlib.sha1()
        h.update(self.alive.tobytes())
        h.update(repr(sorted(self.membership.dropped)).encode())
        h.update("|".join(self.roles).encode())
        return h.digest()

    def _build_exchange(self, train_sets, adopt):
        """Exchange plan of this rank (see parallel/planner.py) in the tuple form ``aggregate`` caches."""
        ep = plan_exchange(train_sets, adopt, self.owner, self.weights, self.rank, self.world, self.local)
        if ep.mode == "allreduce":
            lo, dev = self.local.start, self.device
            plan = idx = None
            if ep.local_sources:
                i32 = dict(dtype=torch.int32, device=dev)
                plan = AggregationPlan(torch.tensor([0, len(ep.local_sources)], **i32),
                                       torch.tensor([j - lo for j in ep.local_sources], **i32),
                                       torch.tensor(ep.local_weights, dtype=torch.float32, device=dev),
                                       torch.tensor([0, 1], **i32), torch.tensor([0], **i32), 1)
            if ep.local_dests:
                idx = torch.tensor([d - lo for d in ep.local_dests], device=dev)
            return ("allreduce", ep.local_sources, ep.local_weights, ep.local_dests, plan, idx)
        return ("rows", ep.send_lists, ep.recv_lists, build_plan(ep.groups), ep.n_recv)

    # ------------------------------------------------------------------ exchange
    def _exchange_rows(self, send_lists, recv_lists, n_recv: int) -> int:
        """Move the needed parameter rows between ranks into ``self.staging``."""
        eng = self.engine
        P = eng.layout.row_stride
        if self.staging is None or self.staging.shape[0] < max(n_recv, 1):
            self.staging = torch.zeros(max(n_recv, 1), P, device=self.device)
        lo = self.local.start
        return self.comm.exchange_rows(eng.params, [[j - lo for j in l] for l in send_lists],
                                       [len(l) for l in recv_lists], self.staging)

    # ------------------------------------------------------------------ the round
    def _apply_faults(self) -> None:
        self.membership.apply_round(self.round)
        self.alive = self.membership.alive().copy()

    def _range(self, name: str):
        """roctx range (visible in rocprofv3 --marker-trace) when P2PFL_TRACE=1."""
        if self.trace and self.device.type == "cuda":
            torch.cuda.nvtx.range_push(name)
            return True
        return False

    def aggregate(self) -> int:
        """Exchange + FedAvg for the current membership/roles. Returns #rows received from peers."""
        eng = self.engine
        cached, all_covered = self._plan()
        if cached is None:
            return 0
        out = eng.alt_slab()
        if not all_covered:
            out.copy_(eng.params)        # rows nobody writes (dead / isolated nodes) keep their model
        return self._run_exchange(cached, out)

    def prepare(self) -> None:
        """Ahead-of-time setup (captured graphs, exchange plan, aggregation buffers) so that the first round
        costs what every round costs; leaves models, optimizer state and counters untouched."""
        if hasattr(self.engine, "prepare"):
            self.engine.prepare(self.epochs)
        self._plan()
        self.engine.alt_slab()

    def _plan(self):
        """(exchange plan, all local rows covered) for the current membership / links / roles.

        Planned once per structure - the Python walk over all nodes of the scenario costs 14 ms for 800 nodes,
        per round and on every rank."""
        key = self._structure_key()
        entry = self._plan_cache.get(key)
        if entry is None:
            train_sets, adopt = self._round_structure()
            if not train_sets:
                entry = (None, True)
            else:
                cached = self._build_exchange(train_sets, adopt)
                if cached[0] == "rows":
                    cached = cached[:3] + (cached[3].to(self.device),) + cached[4:]
                covered = set(train_sets) | set(adopt)
                entry = (cached, all(i in covered for i in self.local))
            self._plan_cache[key] = entry
        return entry

    def _run_exchange(self, cached, out) -> int:
        eng = self.engine
        if cached[0] == "allreduce":
            plan, idx = cached[4], cached[5]       # device-side plan / destination rows, built once per structure
            P = eng.layout.row_stride
            if self._partial is None or self._partial.shape[1] != P:
                self._partial = torch.zeros(1, P, device=self.device)
            partial = self._partial
            if plan is not None:
                run_plan(eng.params, None, partial, plan)      # (the reduce kernel overwrites its destination rows)
            else:
                partial.zero_()
            partial = self.comm.allreduce_partial(partial)
            if idx is not None:
                out[idx] = partial.expand(idx.numel(), -1)
            eng.swap_slabs()
            return self.world - 1
        _, send_lists, recv_lists, plan, n_recv = cached
        got = self._exchange_rows(send_lists, recv_lists, n_recv)
        run_plan(eng.params, self.staging, out, plan)
        eng.swap_slabs()
        return got

    def _transfer_leadership(self) -> None:
        """SDFL: the aggregator hands its role to a random live neighbour (node.py:676-686)."""
        for i in range(self.N):
            if self.roles[i] == Role.AGGREGATOR and self.alive[i]:
                cands = [j for j in self.neighbors(i) if self.alive[j]]
                if cands:
                    j = self.rng.choice(cands)
                    self.roles[i] = Role.TRAINER
                    self.roles[j] = Role.AGGREGATOR
                break

    def run_round(self, evaluate: bool = True, sync_timing: bool = True) -> RoundReport:
        eng = self.engine
        t0 = time.perf_counter()
        phases: Dict[str, float] = {}
        self._apply_faults()

        def tick(name, t_prev):
            if sync_timing:
                eng.synchronize()
            now = time.perf_counter()
            phases[name] = now - t_prev
            return now

        t = t0
        test = None
        if evaluate:                      # evaluation happens BEFORE training (node.py:433-435)
            pushed = self._range("evaluate")
            test = eng.evaluate()
            if pushed:
                torch.cuda.nvtx.range_pop()
            t = tick("evaluate", t)
        lo = self.local.start
        active = torch.tensor([bool(self.alive[i]) and Role.trains(self.roles[i]) for i in self.local])
        pushed = self._range("train")
        fit = eng.fit(self.epochs, active=active)
        if pushed:
            torch.cuda.nvtx.range_pop()
        t = tick("train", t)
        pushed = self._range("exchange+aggregate")
        got = self.aggregate()
        if pushed:
            torch.cuda.nvtx.range_pop()
        t = tick("aggregate", t)
        if self.federation == "SDFL":
            self._transfer_leadership()
        # virtual heartbeats: the clock moves by the wall time of the round (at least one period)
        self.membership.advance(max(time.perf_counter() - t0, self.membership.period))
        self.alive = self.membership.alive().copy()
        n_train = eng.train_x.shape[1]
        rep = RoundReport(self.round, time.perf_counter() - t0, phases,
                          int(active.sum().item()) * n_train * self.epochs,
                          None if test is None else test["loss"], None if test is None else test["accuracy"],
                          fit.get("Train", {}).get("loss") if fit else None, got)
        self.reports.append(rep)
        self.round += 1
        return rep

    def run(self, rounds: int, evaluate: bool = True) -> List[RoundReport]:
        out = [self.run_round(evaluate=evaluate) for _ in range(rounds)]
        if evaluate:                      # one more evaluation at the very end (node.py:663)
            self.final_test = self.engine.evaluate()
        return out
