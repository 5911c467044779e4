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
* a dead node (fault injection) stops training at once but stays in its neighbours' train sets until the
  heartbeat table evicts it ``NODE_TIMEOUT`` later (heartbeater.py:88-101): meanwhile every aggregator that expects
  its model waits ``AGGREGATION_TIMEOUT`` on the (virtual) clock and then averages the subset that did arrive
  (aggregator.py:47-76,181-208) - all those partial aggregations are groups of ONE launch of the reduce kernel.

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
from .aggregation import (AggregationPlan, build_plan, inplace_supported, run_plan, run_plan_inplace,
                          single_group_inplace)
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
    timeouts: int = 0                # aggregators that gave up on a missing contributor (partial aggregation)
    encrypted_bytes: int = 0         # bytes that went through the link cipher in this round


class BatchedFederation:
    def __init__(self, spec: ModelSpec, topology, *, federation: str = "DFL", roles: Optional[Sequence[str]] = None,
                 device="cpu", engine: str = "auto", batch_size: int = 32, epochs: int = 1, seed: int = 0,
                 rank: int = 0, world_size: int = 1, process_group=None, encrypt: bool = False,
                 scenario: str = "scenario", persist_optimizer: bool = False, use_graph: bool = True,
                 overlap: bool = True, heartbeat_period: float = 4.0, node_timeout: float = 20.0,
                 instant_detection: bool = True, precision: Optional[str] = None,
                 aggregation_timeout: float = 60.0, virtual_devices: int = 1,
                 block_sizes: Optional[Sequence[int]] = None, log_steps: bool = False):
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
        self.blocks = partition_nodes(self.N, self.world, block_sizes)   # block_sizes: explicit placement (launch.py)
        self.owner = np.zeros(self.N, dtype=np.int64)
        for r, blk in enumerate(self.blocks):
            self.owner[list(blk)] = r
        self.local = self.blocks[rank]
        self.n_local = len(self.local)
        self.engine: ShardEngine = make_engine(spec, self.n_local, self.device, impl=engine, batch_size=batch_size,
                                               seed=seed, persist_optimizer=persist_optimizer,
                                               client_offset=self.local.start,
                                               **({"use_graph": use_graph, "overlap": overlap, "precision": precision}
                                                  if engine != "eager" else {}),
                                               # per-step Train/* series (snapshots inside the captured graphs)
                                               **({"log_steps": True} if log_steps else {}))
        # membership of ALL nodes of the scenario (every rank applies the same deterministic schedule)
        self.membership = VirtualMembership(self.N, heartbeat_period, node_timeout, instant=instant_detection)
        self.alive = np.ones(self.N, dtype=bool)                # the node's process is up (it evaluates / trains)
        self.believed = np.ones(self.N, dtype=bool)             # its neighbours still count it into their train sets
        self.delivered = np.ones(self.N, dtype=bool)            # its model of this round reaches its neighbours
        self.aggregation_timeout = float(aggregation_timeout)
        self.external_clock = False                             # a FederationHost drives the membership clock itself
        self.pre_aggregate_hook = None                          # callables (fed) -> None set by a FederationHost
        self.post_aggregate_hook = None
        self.weights = np.zeros(self.N, dtype=np.float64)       # training samples per node
        self.round = 0
        self.rng = random.Random(seed)                            # shared by all ranks (same seed)
        self.comm = DeviceExchange(rank, world_size, process_group, self.device, encrypt, scenario)
        # Host-side group (gloo) next to the RCCL one: what the ranks have to AGREE on - the heartbeat table's evictions
        # when they are timing-dependent (NODE_TIMEOUT detection), STOP_LEARNING, which nodes are TCP gateways - is a few
        # bytes per round and never travels on the RCCL stream (see _sync_membership, runtime/host.py)
        self.host_pg = None
        if world_size > 1 and process_group is not None and not instant_detection:
            self.ensure_host_group()
        # False: never collapse the exchange into one all-reduce even when every aggregator shares one train set -
        # a TCP gateway (runtime/host.py) needs the ROWS of its train set staged on its rank
        self.allow_allreduce = True
        self._row_of: Dict[int, int] = {}                      # node id -> row of the source pool of the current plan
        # >= 2: the nodes of this GPU are split into "virtual devices"; rows that cross from one to another take the
        # inter-GPU path of a multi-GPU job (gather -> link cipher -> staging -> decrypt -> reduce over local +
        # staged rows) although they never leave the card - config #4 (encrypted exchange) on a one-GPU box
        self.virtual_devices = int(virtual_devices) if world_size == 1 else 1
        self._vd_cache: Dict[bytes, tuple] = {}
        self.staging: Optional[torch.Tensor] = None
        self._partial: Optional[torch.Tensor] = None
        self._plan_cache: Dict[bytes, tuple] = {}
        self.trace = os.environ.get("P2PFL_TRACE", "0") not in ("", "0")   # roctx ranges around the phases
        self.reports: List[RoundReport] = []

    def ensure_host_group(self):
        """The gloo group of this job's ranks (the process group itself when that is gloo already). Collective: every
        rank calls it at the same point."""
        if self.host_pg is None and self.pg is not None and self.world > 1:
            import torch.distributed as dist
            self.host_pg = self.pg if dist.get_backend(self.pg) == "gloo" else dist.new_group(backend="gloo")
        return self.host_pg

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
        """Resume at ``first_round``: the faults of the rounds already done are applied and the membership clock moves
        as it did when those rounds ran - per replayed round the start faults, the mid-round faults, the round
        structure (which aggregators waited in vain for a dead member) and then one heartbeat period plus
        AGGREGATION_TIMEOUT when somebody timed out, which is what evicts dead nodes - so ``believed`` / evictions, the
        ``late`` tie-break of adopting nodes after ``--resume`` are those of the uninterrupted run (roles, i.e. the
        SDFL leader, come from the checkpoints). No training; the round and per-round shuffle counters continue."""
        for r in range(self.round, first_round):
            self.round = r
            self._apply_faults("start")
            self._apply_faults("mid")
            self._round_structure()
            timed_out = list(self.timed_out)
            for i in timed_out:
                self.membership.note("aggregation_timeout", i)
            self.membership.advance(self.membership.period + (self.aggregation_timeout if timed_out else 0.0))
            self._sync_membership()
        self.round = first_round
        self.engine._perm_calls = first_round

    def schedule_fault(self, round_idx: int, action: str, node: int, peer: Optional[int] = None,
                       phase: str = "start") -> None:
        """Deterministic churn: ``kill`` / ``revive`` / ``pause`` / ``resume`` of a node, ``drop_link`` /
        ``restore_link`` of the edge (node, peer), at the start of a round or in the middle of it (``phase="mid"``:
        after local training, before the exchange); see runtime/membership.py."""
        self.membership.schedule(round_idx, action, node, peer, phase)

    # ------------------------------------------------------------------ per-round structure
    def neighbors(self, i: int) -> List[int]:
        return [int(j) for j in np.nonzero(self.adj[i])[0] if j != i and self.membership.link_up(i, int(j))]

    def train_set(self, i: int) -> List[int]:
        """What node i EXPECTS to aggregate: its direct neighbours that the heartbeat table still lists, plus
        itself (node.py:411-422,537-548)."""
        if not self.alive[i]:
            return []
        return sorted([j for j in self.neighbors(i) if self.believed[j]] + [i])

    def _present(self, j: int) -> bool:
        return bool(self.alive[j] and self.delivered[j])

    def _round_structure(self):
        """(train_sets of aggregating nodes, adopt map of waiting nodes) for the current state. A train set here is
        what ARRIVES: expected members that are down (not yet evicted) or died mid-round are missing, the
        aggregator times out and averages the rest (``self.timed_out`` lists those aggregators)."""
        train_sets: Dict[int, List[int]] = {}
        adopt: Dict[int, int] = {}
        self.timed_out: List[int] = []
        for i in range(self.N):
            if self._present(i) and Role.aggregates(self.roles[i]):
                # contributors: members of the train set that hold a model to contribute
                # (proxy/idle nodes never add their own model, node.py:492-524)
                expected = [j for j in self.train_set(i) if self.roles[j] not in (Role.PROXY, Role.IDLE)]
                ts = [j for j in expected if self._present(j)]
                if len(ts) < len(expected):
                    self.timed_out.append(i)
                train_sets[i] = ts
        for i in range(self.N):
            if self._present(i) and Role.waits(self.roles[i]):
                aggs = [j for j in self.neighbors(i) if self._present(j) and j in train_sets]
                if aggs:
                    # "the first aggregated model that arrives" (node.py:526-531) on the virtual clock: an
                    # aggregator that waits out AGGREGATION_TIMEOUT for a missing member answers last, otherwise
                    # the one that collects the fewest models is done first; the node index breaks ties
                    late = set(self.timed_out)
                    adopt[i] = min(aggs, key=lambda j: (j in late, len(train_sets[j]), j))
        return train_sets, adopt

    def _structure_key(self) -> bytes:
        h = hashlib.sha1()
        h.update(self.alive.tobytes())
        h.update(self.believed.tobytes())
        h.update(self.delivered.tobytes())
        h.update(repr(sorted(self.membership.dropped)).encode())
        h.update("|".join(self.roles).encode())
        h.update(b"1" if self.allow_allreduce else b"0")
        return h.digest()

    def _build_exchange(self, train_sets, adopt):
        """Exchange plan of this rank (see parallel/planner.py) in the tuple form ``aggregate`` caches."""
        ep = plan_exchange(train_sets, adopt, self.owner, self.weights, self.rank, self.world, self.local,
                           allow_allreduce=self.allow_allreduce)
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
        return ("rows", ep.send_lists, ep.recv_lists, build_plan(ep.groups), ep.n_recv, dict(ep.row_of))

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
    def _apply_faults(self, phase: str = "start") -> None:
        before = self.membership.running().copy()
        self.membership.apply_round(self.round, phase)
        self._sync_membership()
        if phase == "start":
            self.delivered[:] = True
        else:
            # a node that went down in the middle of the round trained for nothing: its model never arrives
            self.delivered &= ~(before & ~self.membership.running())

    def _sync_membership(self) -> None:
        self.alive = self.membership.running().copy()
        self.believed = self.membership.believed().copy()
        if self.host_pg is not None and not self.membership.instant:
            # Faults are scheduled per round, so ``alive`` is the same on every rank by construction; WHEN the heartbeat
            # table evicts a silent node depends on each rank's clock (wall time of its rounds / its timer wheel). The
            # exchange plans of all ranks must describe one structure - what a sends to b is what b expects - so the
            # ranks agree here, on the host-side group: a node is evicted everywhere as soon as one rank evicted it.
            import torch.distributed as dist
            t = torch.from_numpy(self.believed.astype(np.uint8))
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.host_pg)
            self.believed = t.numpy().astype(bool)
            late = self.membership.member & ~self.believed
            for i in np.nonzero(late)[0]:
                self.membership.member[i] = False
                self.membership.events.append((self.membership.clock, "evicted", int(i)))

    def _range(self, name: str):
        """roctx range (visible in rocprofv3 --marker-trace) when P2PFL_TRACE=1."""
        if self.trace and self.device.type == "cuda":
            torch.cuda.nvtx.range_push(name)
            return True
        return False

    def aggregate(self) -> int:
        """Exchange + FedAvg for the current membership/roles. Returns #rows received from peers.

        ``pre_aggregate_hook`` (a FederationHost) runs BETWEEN the two: the rows of other ranks are staged by then, so
        a TCP gateway can form the partial aggregate of its whole train set (local + staged rows) while the reduce of
        the slab waits for the gateway's own aggregation."""
        eng = self.engine
        hook = self.pre_aggregate_hook if self.pre_aggregate_hook is not None else (lambda fed: None)
        if self.virtual_devices > 1:
            hook(self)
            return self._aggregate_virtual_devices()
        cached, all_covered = self._plan()
        lo = self.local.start
        self._row_of = cached[5] if (cached is not None and cached[0] == "rows") else {j: j - lo for j in self.local}
        if cached is None:
            hook(self)
            return 0
        if cached[0] == "rows":
            _, send_lists, recv_lists, plan, n_recv = cached[:5]
            got = self._exchange_rows(send_lists, recv_lists, n_recv)
            hook(self)
            if inplace_supported(eng.params, n_recv) or single_group_inplace(eng.params, plan):
                # one slab: the reduce reads every row once and overwrites the destinations in place (no second slab,
                # no copy back into the slab the captured graphs point at; uncovered rows keep their model by themselves)
                if run_plan_inplace(eng.params, self.staging, plan, n_recv):
                    return got
                raise RuntimeError("in-place FedAvg declined a plan it had accepted")
            out = eng.alt_slab()
            if not all_covered:
                out.copy_(eng.params)        # rows nobody writes (dead / isolated nodes) keep their model
            run_plan(eng.params, self.staging, out, plan)
            eng.swap_slabs()
            return got
        hook(self)
        out = eng.alt_slab()
        if not all_covered:
            out.copy_(eng.params)
        return self._run_exchange(cached, out)

    def _aggregate_virtual_devices(self) -> int:
        """The exchange of a V-GPU job played on one card (see ``virtual_devices``): per virtual device one reduce
        over its own rows + the staged rows of the others, every staged row through the (src, dst) link cipher
        when ``encrypt`` is on."""
        eng, V = self.engine, self.virtual_devices
        key = self._structure_key()
        entry = self._vd_cache.get(key)
        if entry is None:
            train_sets, adopt = self._round_structure()
            blocks = partition_nodes(self.N, V)
            vowner = np.zeros(self.N, dtype=np.int64)
            for v, blk in enumerate(blocks):
                vowner[list(blk)] = v
            plans = []
            for v in range(V):
                ep = plan_exchange(train_sets, adopt, vowner, self.weights, v, V, blocks[v], allow_allreduce=False)
                idx = [torch.tensor(l, dtype=torch.long, device=self.device) if l else None for l in ep.recv_lists]
                plans.append((blocks[v], idx, build_plan(ep.groups).to(self.device), ep.n_recv))
            covered = set(train_sets) | set(adopt)
            entry = (plans if train_sets else None, all(i in covered for i in range(self.N)), list(self.timed_out))
            self._vd_cache[key] = entry
        plans, all_covered, self.timed_out = entry[0], entry[1], list(entry[2])
        if plans is None:
            return 0
        out = eng.alt_slab()
        if not all_covered:
            out.copy_(eng.params)
        P = eng.layout.row_stride
        need = max(max(p[3] for p in plans), 1)
        if self.staging is None or self.staging.shape[0] < need:
            self.staging = torch.zeros(need, P, device=self.device)
        moved = 0
        for v, (blk, idx, plan, n_recv) in enumerate(plans):
            off = 0
            for r in range(V):
                if idx[r] is None:
                    continue
                n = idx[r].numel()
                rows = eng.params.index_select(0, idx[r])                    # what device r sends to device v
                if self.encrypt:
                    wire = self.comm.crypt(rows, v, decrypt=False, local=r).clone()       # on the link: ciphertext
                    self.comm.crypt(wire, r, decrypt=True, out=self.staging[off:off + n], local=v)
                else:
                    self.staging[off:off + n] = rows
                off += n
            moved += off
            run_plan(eng.params[blk.start:blk.stop], self.staging, out[blk.start:blk.stop], plan)
        self.comm.bytes_sent += moved * P * 4
        eng.swap_slabs()
        return moved

    def prepare(self) -> None:
        """Ahead-of-time setup (captured graphs, exchange plan, aggregation buffers) so that the first round
        costs what every round costs; leaves models, optimizer state and counters untouched."""
        if hasattr(self.engine, "prepare"):
            self.engine.prepare(self.epochs)
        cached, _ = self._plan()
        if self.virtual_devices > 1 or cached is None or not (cached[0] == "rows" and (
                inplace_supported(self.engine.params, cached[4]) or single_group_inplace(self.engine.params, cached[3]))):
            self.engine.alt_slab()       # (the in-place reduce needs no second slab)

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
            entry = entry + (list(self.timed_out),)
            self._plan_cache[key] = entry
        self.timed_out = list(entry[2])
        return entry[:2]

    def _run_exchange(self, cached, out) -> int:
        """The all-reduce form of the exchange (every aggregator shares one train set)."""
        eng = self.engine
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
        enc0 = self.comm.bytes_encrypted
        self._apply_faults("start")

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
        self._apply_faults("mid")             # nodes that die now have trained, but their model never arrives
        pushed = self._range("exchange+aggregate")
        got = self.aggregate()
        if pushed:
            torch.cuda.nvtx.range_pop()
        if self.post_aggregate_hook is not None:
            self.post_aggregate_hook(self)
        t = tick("aggregate", t)
        if self.federation == "SDFL":
            self._transfer_leadership()
        # Virtual heartbeats: the clock moves by the wall time of the round (at least one period) - plus
        # AGGREGATION_TIMEOUT when some aggregator waited in vain for a member that is down but not yet evicted
        # (aggregator.py:47-76). That wait is what lets NODE_TIMEOUT expire and the eviction happen.
        timed_out = list(getattr(self, "timed_out", []))
        for i in timed_out:
            self.membership.note("aggregation_timeout", i)
        if not self.external_clock:
            # (several ranks: one heartbeat period per round whatever the wall time, so every rank's table moves alike)
            wall = self.membership.period if self.world > 1 else max(time.perf_counter() - t0, self.membership.period)
            self.membership.advance(wall + (self.aggregation_timeout if timed_out else 0.0))
        self._sync_membership()
        n_train = eng.train_x.shape[1]
        rep = RoundReport(self.round, time.perf_counter() - t0, phases,
                          int(active.sum().item()) * n_train * self.epochs,
                          None if test is None else test["loss"], None if test is None else test["accuracy"],
                          fit.get("Train", {}).get("loss") if fit else None, got, len(timed_out),
                          self.comm.bytes_encrypted - enc0)
        self.reports.append(rep)
        self.round += 1
        return rep

    def run(self, rounds: int, evaluate: bool = True) -> List[RoundReport]:
        out = [self.run_round(evaluate=evaluate) for _ in range(rounds)]
        if evaluate:                      # one more evaluation at the very end (node.py:663)
            self.final_test = self.engine.evaluate()
        return out
